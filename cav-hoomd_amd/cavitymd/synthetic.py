"""Synthetic inputs for the five benchmark/parity configurations (BASELINE.json ``configs``).

The reference's own input, ``examples/init-0.gsd``, is missing from its checkout, so config 1 is a
stand-in with the schema its driver expects (500 particles of types 'O'/'N' in diatomics, charges, and a
photon of type 'L' = typeid 2, charge 0, appended LAST exactly as ``create_cavity_particle`` does,
examples/05_advanced_run.py:453-537 of the reference).  Generated on the spot from a seed with numpy's
PCG64; nothing here is shipped as data.

Every generator returns a dict in snapshot form:
    position (N,3) f64 wrapped into [-L/2, L/2), typeid (N,) i32, charge (N,) f64, image (N,3) i32,
    types ['O','N','L'], box (Lx,Ly,Lz), L_typeid, params {omegac, couplstr, phmass}, name, seed
"""
from __future__ import annotations

import numpy as np

from .utils import PhysicalConstants

TYPES = ["O", "N", "L"]
L_TYPEID = 2
DEFAULT_COUPLING = 1e-3
DEFAULT_FREQ_CM = 2000.0
DEFAULT_TEMPERATURE = 100.0
BOND_OO = 2.2817  # bohr, examples/05_advanced_run.py:568
BOND_NN = 2.0744  # bohr, examples/05_advanced_run.py:569


def default_params(couplstr: float = DEFAULT_COUPLING, freq_cm: float = DEFAULT_FREQ_CM, phmass: float = 1.0) -> dict:
    return {"omegac": PhysicalConstants.omegac_from_wavenumber(freq_cm), "couplstr": float(couplstr),
            "phmass": float(phmass)}


def wrap(r: np.ndarray, L: np.ndarray):
    """Wrapped position and image flags such that r = wrapped + image * L (as examples/05_advanced_run.py:487-493)."""
    img = np.floor((r + L / 2) / L)
    return r - img * L, img.astype(np.int32)


def _photon_position(rng, dipole, params, finite_q: bool, kT: float):
    """Initial photon position, examples/05_advanced_run.py:464-485: thermal Gaussian about 0 (q=0 start) or
    about -d g / omegac^2 with z zeroed (finite-q start)."""
    omegac, g = params["omegac"], params["couplstr"]
    sigma = np.sqrt(kT / omegac**2)
    if finite_q:
        loc = -dipole * g / omegac**2
        loc[2] = 0.0
    else:
        loc = np.zeros(3)
    return rng.normal(loc=loc, scale=sigma, size=3) if g != 0.0 else loc


def _append_photon(rng, position, typeid, charge, image, box, params, finite_q, kT):
    unwrapped = position + image * box[None, :]
    dipole = np.einsum("i,ij->j", charge, unwrapped)
    q = _photon_position(rng, dipole, params, finite_q, kT)
    qw, qi = wrap(q, box)
    position = np.vstack([position, qw[None, :]])
    image = np.vstack([image, qi[None, :]]).astype(np.int32)
    typeid = np.append(typeid, L_TYPEID).astype(np.int32)
    charge = np.append(charge, 0.0)
    return position, typeid, charge, image


def diatomic_box(n_molecular: int, seed: int, box_length: float | None = None, finite_q: bool = False,
                 image_range: int = 1, params: dict | None = None, name: str = "diatomic") -> dict:
    """Neutral O2-like / N2-like diatomics (+delta, -delta per molecule) plus the photon.  N = n_molecular + 1."""
    assert n_molecular % 2 == 0
    rng = np.random.default_rng(seed)
    params = params or default_params()
    kT = PhysicalConstants.KB_HARTREE_PER_K * DEFAULT_TEMPERATURE
    n_mol = n_molecular // 2
    if box_length is None:
        box_length = (n_molecular / 0.01)**(1.0 / 3.0)
    box = np.array([box_length] * 3)
    centre = rng.uniform(-box_length / 2, box_length / 2, size=(n_mol, 3))
    direction = rng.normal(size=(n_mol, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    species = (np.arange(n_mol) % 2).astype(np.int32)  # 0 = O-O, 1 = N-N
    bond = np.where(species == 0, BOND_OO, BOND_NN)[:, None]
    mol_image = rng.integers(-image_range, image_range + 1, size=(n_mol, 3)).astype(np.float64)
    r_a = centre + 0.5 * bond * direction + mol_image * box[None, :]
    r_b = centre - 0.5 * bond * direction + mol_image * box[None, :]
    r = np.empty((n_molecular, 3))
    r[0::2] = r_a
    r[1::2] = r_b
    delta = rng.uniform(0.1, 0.5, size=n_mol)
    charge = np.empty(n_molecular)
    charge[0::2] = delta
    charge[1::2] = -delta
    typeid = np.repeat(species, 2).astype(np.int32)
    position, image = wrap(r, box)
    position, typeid, charge, image = _append_photon(rng, position, typeid, charge, image, box, params, finite_q, kT)
    return {"name": name, "seed": seed, "position": position, "typeid": typeid, "charge": charge, "image": image,
            "types": list(TYPES), "box": tuple(box), "L_typeid": L_TYPEID, "params": params, "finite_q": finite_q}


def random_charged_box(n_molecular: int, seed: int, finite_q: bool = False, image_range: int = 2,
                       params: dict | None = None, name: str = "random_charged_box") -> dict:
    """Uniform positions, charges U(-1,1) with the mean removed, images U{-2..2}, typeid = i mod 2, + photon."""
    rng = np.random.default_rng(seed)
    params = params or default_params()
    kT = PhysicalConstants.KB_HARTREE_PER_K * DEFAULT_TEMPERATURE
    box_length = (n_molecular / 0.01)**(1.0 / 3.0)
    box = np.array([box_length] * 3)
    position = rng.uniform(-box_length / 2, box_length / 2, size=(n_molecular, 3))
    charge = rng.uniform(-1.0, 1.0, size=n_molecular)
    charge -= charge.mean()
    image = rng.integers(-image_range, image_range + 1, size=(n_molecular, 3)).astype(np.int32)
    typeid = (np.arange(n_molecular) % 2).astype(np.int32)
    position, typeid, charge, image = _append_photon(rng, position, typeid, charge, image, box, params, finite_q, kT)
    return {"name": name, "seed": seed, "position": position, "typeid": typeid, "charge": charge, "image": image,
            "types": list(TYPES), "box": tuple(box), "L_typeid": L_TYPEID, "params": params, "finite_q": finite_q}


# ---- the five configurations of BASELINE.json / SURVEY.md 8(d) ---------------------------------------
def config1(seed: int = 1) -> dict:
    """Stand-in for examples/init-0.gsd: 125 O-O + 125 N-N diatomics in a 40-bohr box, N = 501."""
    return diatomic_box(500, seed, box_length=40.0, finite_q=False, image_range=1, name="config1_init0_standin")


def config2(seed: int = 2, n_molecular: int = 100_000) -> dict:
    return random_charged_box(n_molecular, seed, finite_q=False, name="config2_random_1e5")


def config3(seed: int = 3, n_molecular: int = 1_000_000) -> dict:
    """1e6 particles, finite-q photon start: the worst-case cancellation input."""
    return diatomic_box(n_molecular, seed, finite_q=True, image_range=1, name="config3_finiteq_1e6")


def config4(seed: int = 4, n_molecular: int = 10_000_000) -> dict:
    return random_charged_box(n_molecular, seed, finite_q=False, name="config4_random_1e7")


def config5_replica(rank: int, n_molecular: int = 1_000_000) -> dict:
    """Replica `rank` of config 5: config 3 with seed rank + 1 (seeds 1-8 across 8 GPUs)."""
    cfg = config3(seed=rank + 1, n_molecular=n_molecular)
    cfg["name"] = f"config5_replica{rank}_seed{rank + 1}"
    return cfg


def perturb(cfg: dict, step_seed: int, amplitude: float = 1e-3) -> dict:
    """One step of the pseudo-trajectory that stands in for the integrator + thermostats: every particle
    (photon included) moves by amplitude * N(0,1) and is re-wrapped.  Returns a new config dict."""
    rng = np.random.default_rng([int(cfg["seed"]), int(step_seed)])
    box = np.asarray(cfg["box"])
    r = cfg["position"] + cfg["image"] * box[None, :]
    r = r + amplitude * rng.standard_normal(r.shape)
    position, image = wrap(r, box)
    out = dict(cfg)
    out["position"], out["image"] = position, image
    return out
