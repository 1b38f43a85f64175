#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_golden.py`); the GPU box never sees /root/reference.

Two kinds of fixture, kept apart because they pin different things:

  utils_golden.json     REFERENCE-GENERATED.  Produced by importing the reference's own
                        src/cavitymd/utils.py by file path (the only reference module that loads
                        without HOOMD-blue; numpy only) and recording its constants, its
                        unit-conversion results and its unwrap_positions() outputs on seeded inputs.
                        Also records the two scalars the reference's notebook prints for 2000 cm^-1
                        (examples/05_advanced_run.ipynb:669).  These pin the unwrap convention and
                        K = phmass * omegac^2 of our oracle and of the product.

  kat_golden.json       HAND-DERIVED known answers of the formulas at src/CavityForceCompute.cc:174-207
                        on inputs chosen so that every intermediate is exactly representable.
  config1_oracle.npz    ORACLE-GENERATED regression vectors (config 1, N = 501): inputs + the outputs of
                        oracle/cavity_ref.c.  They are NOT reference outputs (the reference cannot run
                        here); they freeze the oracle so the GPU parity tests also compare against
                        committed numbers.
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
REF_UTILS = "/root/reference/src/cavitymd/utils.py"


def load_reference_utils():
    spec = importlib.util.spec_from_file_location("ref_cavitymd_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_utils_golden():
    ref = load_reference_utils()
    PC = ref.PhysicalConstants
    rng = np.random.default_rng(20250704)
    cases = []
    for n, L in ((1, (10.0, 10.0, 10.0)), (7, (40.0, 40.0, 40.0)), (64, (215.44346900318845, 215.44346900318845,
                                                                           215.44346900318845)),
                 (33, (12.5, 31.25, 7.0))):
        pos = rng.uniform(-0.5, 0.5, size=(n, 3)) * np.asarray(L)
        img = rng.integers(-3, 4, size=(n, 3)).astype(np.int32)
        out = ref.unwrap_positions(pos, img, np.asarray(L))
        cases.append({"positions": pos.tolist(), "images": img.tolist(), "box": list(L), "unwrapped": out.tolist()})
    # the survey's spot check
    spot = ref.unwrap_positions([[1, 2, 3]], [[1, 0, -1]], [10, 10, 10]).tolist()
    constants = {k: getattr(PC, k) for k in ("HARTREE_TO_CM_MINUS1", "KB_HARTREE_PER_K", "ENERGY_JOULES",
                                             "LENGTH_METERS", "MASS_KG", "TIME_SECONDS", "TIME_PS_CONVERSION")}
    conversions = {
        "ps_to_atomic_units": [[t, PC.ps_to_atomic_units(t)] for t in (0.001, 1.0, 5.0, 1000.0)],
        "atomic_units_to_ps": [[t, PC.atomic_units_to_ps(t)] for t in (1.0, 41341.37, 1e6)],
        "gamma_from_tau_ps": [[t, PC.gamma_from_tau_ps(t)] for t in (0.1, 1.0, 5.0)],
    }
    omegac = 2000.0 / PC.HARTREE_TO_CM_MINUS1
    data = {
        "generated_by": "tests/golden/make_golden.py from the reference's src/cavitymd/utils.py",
        "constants": constants,
        "conversions": conversions,
        "unwrap_cases": cases,
        "unwrap_spot": spot,
        "omegac_2000cm": omegac,
        "K_2000cm_phmass1": 1.0 * omegac * omegac,
        # what the reference's notebook prints (6 significant digits), examples/05_advanced_run.ipynb:669
        "notebook_printed": {"omegac": "0.00911267", "K": "8.30408e-05"},
    }
    with open(os.path.join(HERE, "utils_golden.json"), "w") as f:
        json.dump(data, f, indent=1)
    print("wrote utils_golden.json")


def make_kat_golden():
    """Closed-form cases; all numbers are dyadic so every product/sum below is exact in fp64."""
    cases = []
    # case A: two charges + photon, no images.  g = 0.5, omegac = 2, phmass = 0.25 -> K = 1
    #   r1 = (1, 2, 3) c1 = +1 ; r2 = (-3, 0.5, 1) c2 = -0.5 ; photon q = (0.25, -0.5, 2)
    #   d = (1 + 1.5, 2 - 0.25, 3 - 0.5) = (2.5, 1.75, 2.5)
    #   E_h = 0.5*1*(0.0625 + 0.25 + 4) = 2.15625 ; E_c = 0.5*(2.5*0.25 + 1.75*-0.5) = 0.5*(-0.25) = -0.125
    #   E_d = 0.5*(0.25/1)*(6.25 + 3.0625) = 0.125*9.3125 = 1.1640625
    #   Dq = (0.25 + 0.5*2.5, -0.5 + 0.5*1.75) = (1.5, 0.375)
    #   F1 = -0.5*1*Dq = (-0.75, -0.1875, 0) ; F2 = -0.5*(-0.5)*Dq = (0.375, 0.09375, 0)
    #   F_L = (-1*0.25 - 0.5*2.5, -1*-0.5 - 0.5*1.75, -1*2) = (-1.5, -0.375, -2)
    cases.append({
        "name": "two_charges_no_images",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[1, 2, 3], [-3, 0.5, 1], [0.25, -0.5, 2]], "typeid": [0, 1, 2], "charge": [1.0, -0.5, 0.0],
        "image": [[0, 0, 0], [0, 0, 0], [0, 0, 0]], "box": [16.0, 16.0, 16.0], "L_typeid": 2,
        "dipole": [2.5, 1.75, 2.5], "energies": [2.15625, -0.125, 1.1640625], "photon_idx": 2,
        "force": [[-0.75, -0.1875, 0, 0], [0.375, 0.09375, 0, 0], [-1.5, -0.375, -2, 0]],
    })
    # case B: same unwrapped geometry expressed through image flags (box 16): r1 = (1-16, 2, 3+32) img (1,0,-2) ...
    cases.append({
        "name": "two_charges_with_images",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[-15, 2, 35], [-3, -15.5, 1], [0.25, -0.5, -14]], "typeid": [0, 1, 2],
        "charge": [1.0, -0.5, 0.0],
        "image": [[1, 0, -2], [0, 1, 0], [0, 0, 1]], "box": [16.0, 16.0, 16.0], "L_typeid": 2,
        "dipole": [2.5, 1.75, 2.5], "energies": [2.15625, -0.125, 1.1640625], "photon_idx": 2,
        "force": [[-0.75, -0.1875, 0, 0], [0.375, 0.09375, 0, 0], [-1.5, -0.375, -2, 0]],
    })
    # case C: photon FIRST, carries a charge (must still be excluded from d), type id 0 is 'L'
    cases.append({
        "name": "photon_first_charged",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[0.25, -0.5, 2], [1, 2, 3], [-3, 0.5, 1]], "typeid": [0, 1, 2], "charge": [4.0, 1.0, -0.5],
        "image": [[0, 0, 0], [0, 0, 0], [0, 0, 0]], "box": [16.0, 16.0, 16.0], "L_typeid": 0,
        "dipole": [2.5, 1.75, 2.5], "energies": [2.15625, -0.125, 1.1640625], "photon_idx": 0,
        "force": [[-1.5, -0.375, -2, 0], [-0.75, -0.1875, 0, 0], [0.375, 0.09375, 0, 0]],
    })
    # case D: no particle of type L -> zeros everywhere (src/CavityForceCompute.cc:148-156)
    cases.append({
        "name": "no_photon",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[1, 2, 3], [-3, 0.5, 1]], "typeid": [0, 1], "charge": [1.0, -0.5],
        "image": [[0, 0, 0], [0, 0, 0]], "box": [16.0, 16.0, 16.0], "L_typeid": 2,
        "dipole": [0.0, 0.0, 0.0], "energies": [0.0, 0.0, 0.0], "photon_idx": -1,
        "force": [[0, 0, 0, 0], [0, 0, 0, 0]],
    })
    # case E: photon only: d = 0, E_h = 0.5 K q.q, F_L = -K q
    cases.append({
        "name": "photon_only",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[0.25, -0.5, 2]], "typeid": [2], "charge": [0.0], "image": [[0, 0, 0]],
        "box": [16.0, 16.0, 16.0], "L_typeid": 2,
        "dipole": [0.0, 0.0, 0.0], "energies": [2.15625, 0.0, 0.0], "photon_idx": 0,
        "force": [[-0.25, 0.5, -2, 0]],
    })
    # case F: two L-typed particles: the first is the photon, the second is INCLUDED in d (only photon_idx is
    # skipped, src/CavityForceCompute.cc:122) but gets NO molecular force (type test at :191).
    #   particles: photon q=(0.25,-0.5,2) c=0 ; mol r=(1,2,3) c=1 ; second L r=(2,2,2) c=0.5
    #   d = (1 + 1, 2 + 1, 3 + 1) = (2, 3, 4); Dq = (0.25 + 0.5*2, -0.5 + 0.5*3) = (1.25, 1.0)
    #   E_c = 0.5*(2*0.25 + 3*-0.5) = -0.5 ; E_d = 0.125*(4+9) = 1.625
    #   F_mol = -0.5*1*Dq = (-0.625, -0.5, 0); F_L2 = 0 ; F_L = (-0.25-1, 0.5-1.5, -2) = (-1.25, -1.0, -2)
    cases.append({
        "name": "two_L_typed",
        "omegac": 2.0, "couplstr": 0.5, "phmass": 0.25, "K": 1.0,
        "position": [[0.25, -0.5, 2], [1, 2, 3], [2, 2, 2]], "typeid": [2, 0, 2], "charge": [0.0, 1.0, 0.5],
        "image": [[0, 0, 0], [0, 0, 0], [0, 0, 0]], "box": [16.0, 16.0, 16.0], "L_typeid": 2,
        "dipole": [2.0, 3.0, 4.0], "energies": [2.15625, -0.5, 1.625], "photon_idx": 0,
        "force": [[-1.25, -1.0, -2, 0], [-0.625, -0.5, 0, 0], [0, 0, 0, 0]],
    })
    with open(os.path.join(HERE, "kat_golden.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden.py (hand-derived, see comments there)", "cases": cases},
                  f, indent=1)
    print("wrote kat_golden.json")


def make_config1_oracle():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "cav-hoomd_amd"))
    import oracle
    from cavitymd import synthetic  # host-side generator only (numpy); no GPU code runs here
    cfg = synthetic.config1(seed=1)
    o = oracle.RefOracle()
    p = o.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
    pos4 = oracle.pack_pos(cfg["position"], cfg["typeid"])
    out = o.compute(pos4, cfg["charge"], cfg["image"], cfg["box"], cfg["L_typeid"], p)
    hi, lo = o.dipole_exact(pos4, cfg["charge"], cfg["image"], cfg["box"], out["photon_idx"])
    np.savez_compressed(os.path.join(HERE, "config1_oracle.npz"), position=cfg["position"], typeid=cfg["typeid"],
                        charge=cfg["charge"], image=cfg["image"], box=np.asarray(cfg["box"]),
                        L_typeid=np.int32(cfg["L_typeid"]),
                        params=np.array([p["omegac"], p["couplstr"], p["K"], p["phmass"]]), force=out["force"],
                        energies=out["energies"], dipole=out["dipole"], photon_idx=np.int32(out["photon_idx"]),
                        dipole_exact_hi=hi, dipole_exact_lo=lo)
    print("wrote config1_oracle.npz")


if __name__ == "__main__":
    make_utils_golden()
    make_kat_golden()
    make_config1_oracle()
