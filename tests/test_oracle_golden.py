"""Pins the CPU oracle (oracle/cavity_ref.c + oracle/numpy_mirror.py).  No GPU.

What the reference offers for this path is: no golden vectors, no KATs.  So the
oracle is pinned by (1) outputs of the reference's own utils.py (unwrap convention, unit constants),
(2) the scalars the reference's notebook prints, (3) hand-derived closed-form cases of the formulas at
src/CavityForceCompute.cc:174-207, (4) F = -dH/dx, (5) exactly rounded sums, (6) an independent numpy
restatement that must agree bit for bit.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import numpy_mirror as nm


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _kat_arrays(oracle_mod, case):
    pos = np.array(case["position"], dtype=np.float64)
    tid = np.array(case["typeid"], dtype=np.int32)
    return (oracle_mod.pack_pos(pos, tid), np.array(case["charge"], dtype=np.float64),
            np.array(case["image"], dtype=np.int32).reshape(-1, 3), case["box"])


# ---- (1) reference-generated fixtures -------------------------------------------------------------------
def test_unwrap_matches_reference_utils(ref, oracle_mod, golden_dir):
    g = _load(golden_dir, "utils_golden.json")
    for case in g["unwrap_cases"]:
        pos = np.array(case["positions"])
        img = np.array(case["images"], dtype=np.int32)
        pos4 = oracle_mod.pack_pos(pos, np.zeros(len(pos), dtype=np.int32))
        got = ref.unwrap(pos4, img, case["box"])
        assert np.array_equal(got, np.array(case["unwrapped"])), "oracle unwrap differs from the reference's utils.py"
        assert np.array_equal(nm.unwrap(pos4, img, case["box"]), np.array(case["unwrapped"]))
    assert g["unwrap_spot"] == [[11, 2, -7]]


def test_K_matches_reference_constants_and_notebook(ref, golden_dir):
    g = _load(golden_dir, "utils_golden.json")
    p = ref.make_params(g["omegac_2000cm"], 1e-3, 1.0)
    assert p["K"] == g["K_2000cm_phmass1"]
    # the reference's notebook prints omegac=0.00911267, K=8.30408e-05 (examples/05_advanced_run.ipynb:669)
    assert f"{p['omegac']:.8f}" == g["notebook_printed"]["omegac"]
    assert f"{p['K']:.5e}" == g["notebook_printed"]["K"]
    # phmass enters linearly, left-to-right association
    p2 = ref.make_params(0.3, 1e-3, 7.0)
    assert p2["K"] == 7.0 * 0.3 * 0.3


def test_layout_sizes(ref):
    assert ref.layout_sizes() == (32, 12, 32, 24)


# ---- (3) closed-form known answers ---------------------------------------------------------------------------
def test_known_answers(ref, ref_o3, oracle_mod, golden_dir):
    kat = _load(golden_dir, "kat_golden.json")
    for case in kat["cases"]:
        pos4, charge, image, box = _kat_arrays(oracle_mod, case)
        for impl in (ref, ref_o3):
            p = impl.make_params(case["omegac"], case["couplstr"], case["phmass"])
            assert p["K"] == case["K"]
            out = impl.compute(pos4, charge, image, box, case["L_typeid"], p)
            assert out["photon_idx"] == case["photon_idx"], case["name"]
            assert np.array_equal(out["dipole"], np.array(case["dipole"], dtype=float)), case["name"]
            assert np.array_equal(out["energies"], np.array(case["energies"], dtype=float)), case["name"]
            assert np.array_equal(out["force"], np.array(case["force"], dtype=float)), case["name"]
        m = nm.compute(pos4, charge, image, box, case["L_typeid"], p)
        assert np.array_equal(m["force"], np.array(case["force"], dtype=float)), case["name"]
        assert np.array_equal(m["energies"], np.array(case["energies"], dtype=float)), case["name"]


def test_type_tag_ignores_high_word(ref, oracle_mod):
    """HOOMD's __scalar_as_int reads only the low 4 bytes of pos.w: garbage in the high word must not matter."""
    pos = np.array([[1.0, 2, 3], [0.25, -0.5, 2]])
    pos4 = oracle_mod.pack_pos(pos, np.array([0, 2]))
    w = pos4[:, 3].view(np.uint64)
    w |= np.uint64(0xDEADBEEF) << np.uint64(32)
    assert ref.find_photon(pos4, 2) == 1
    assert ref.find_photon(pos4, 0) == 0
    assert ref.find_photon(pos4, 5) == -1


# ---- (6) two independent restatements agree bit for bit ------------------------------------------------------------
@pytest.mark.parametrize("seed,n,photon_at", [(0, 1, 0), (1, 2, 1), (2, 17, 5), (3, 501, 500), (4, 1000, 0), (5, 4097, 4096)])
def test_c_oracle_equals_numpy_mirror(ref, ref_o3, oracle_mod, seed, n, photon_at):
    rng = np.random.default_rng(seed)
    L = (31.0, 17.5, 23.25)
    pos = rng.uniform(-0.5, 0.5, (n, 3)) * np.asarray(L)
    tid = rng.integers(0, 2, n).astype(np.int32)
    tid[photon_at] = 2
    charge = rng.uniform(-1, 1, n)
    image = rng.integers(-3, 4, (n, 3)).astype(np.int32)
    pos4 = oracle_mod.pack_pos(pos, tid)
    p = ref.make_params(0.0091, 1e-3, 1.0)
    a = ref.compute(pos4, charge, image, L, 2, p)
    b = nm.compute(pos4, charge, image, L, 2, p, "sequential")
    c = ref_o3.compute(pos4, charge, image, L, 2, p)
    for k in ("force", "energies", "dipole"):
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k + " (-O3 must not change results)"
    assert a["photon_idx"] == b["photon_idx"] == photon_at


def test_golden_config1_regression(ref, oracle_mod, golden_dir):
    g = np.load(os.path.join(golden_dir, "config1_oracle.npz"))
    pos4 = oracle_mod.pack_pos(g["position"], g["typeid"])
    prm = dict(zip(("omegac", "couplstr", "K", "phmass"), g["params"].tolist()))
    out = ref.compute(pos4, g["charge"], g["image"], g["box"], int(g["L_typeid"]), prm)
    assert np.array_equal(out["force"], g["force"])
    assert np.array_equal(out["energies"], g["energies"])
    assert np.array_equal(out["dipole"], g["dipole"])
    assert out["photon_idx"] == int(g["photon_idx"]) == 500


def test_config1_generator_is_reproducible(golden_dir):
    """The synthetic stand-in for init-0.gsd is regenerated from its seed, not shipped: same seed, same bits."""
    from cavitymd import synthetic
    g = np.load(os.path.join(golden_dir, "config1_oracle.npz"))
    cfg = synthetic.config1(seed=1)
    assert np.array_equal(cfg["position"], g["position"])
    assert np.array_equal(cfg["charge"], g["charge"])
    assert np.array_equal(cfg["image"], g["image"])
    assert np.array_equal(cfg["typeid"], g["typeid"])
    assert cfg["typeid"][-1] == 2 and cfg["charge"][-1] == 0.0 and len(cfg["charge"]) == 501


# ---- (5) exactly rounded sums -------------------------------------------------------------------------------------------
def test_exact_dipole_matches_fsum_and_rational(ref, oracle_mod):
    rng = np.random.default_rng(11)
    n = 300
    L = (40.0, 40.0, 40.0)
    pos = rng.uniform(-20, 20, (n, 3))
    tid = np.zeros(n, dtype=np.int32)
    tid[-1] = 2
    charge = rng.uniform(-1, 1, n)
    image = rng.integers(-2, 3, (n, 3)).astype(np.int32)
    pos4 = oracle_mod.pack_pos(pos, tid)
    hi, lo = ref.dipole_exact(pos4, charge, image, L, n - 1)
    t = nm.terms(pos4, charge, image, L)[:-1]
    fs = np.array([math.fsum(t[:, k].tolist()) for k in range(3)])
    assert np.array_equal(hi, fs), "double-double sum must round to the fsum result"
    assert np.all(np.abs(lo) <= np.spacing(np.abs(hi)))
    # against exact rational arithmetic (no rounding of the addends either): within a few ulp of sum|t|
    exact = nm.dipole_rational(pos4, charge, image, L, n - 1)
    scale = np.abs(t).sum(axis=0)
    for k in range(3):
        assert abs(float(exact[k]) - hi[k]) <= 4 * n * np.finfo(float).eps * scale[k] / n + 1e-300
    # the reference's sequential order sits within the classical bound of the exact value
    seq = ref.compute(pos4, charge, image, L, 2, ref.make_params(0.0091, 1e-3, 1.0))["dipole"]
    assert np.all(np.abs(seq - hi) <= n * np.finfo(float).eps * scale)


# ---- (4) forces are minus the gradient of the Hamiltonian ------------------------------------------------------------------
def test_forces_are_gradient_of_hamiltonian(ref, oracle_mod):
    rng = np.random.default_rng(5)
    n = 12
    L = (50.0, 50.0, 50.0)
    pos = rng.uniform(-5, 5, (n, 3))
    tid = (np.arange(n) % 2).astype(np.int32)
    tid[4] = 2
    charge = rng.uniform(-1, 1, n)
    charge[4] = 0.0
    image = rng.integers(-1, 2, (n, 3)).astype(np.int32)
    # strong coupling so every term matters
    p = ref.make_params(0.7, 0.3, 1.3)
    pos4 = oracle_mod.pack_pos(pos, tid)
    F = ref.compute(pos4, charge, image, L, 2, p)["force"]
    h = 1e-4
    for i in range(n):
        for k in range(3):
            pp = pos4.copy()
            pp[i, k] += h
            pm = pos4.copy()
            pm[i, k] -= h
            grad = (nm.hamiltonian(pp, charge, image, L, 2, p) - nm.hamiltonian(pm, charge, image, L, 2, p)) / (2 * h)
            expect = -grad
            if i != 4 and k == 2:
                # the reference zeroes the molecular z force: H couples only d_xy (src/CavityForceCompute.cc:198)
                assert F[i, 2] == 0.0
                assert abs(expect) < 1e-9
            else:
                assert F[i, k] == pytest.approx(expect, rel=1e-7, abs=1e-9), (i, k)
    assert np.all(F[:, 3] == 0.0)


def test_forces_match_the_documented_equations_of_motion(ref, oracle_mod):
    """A pin the reference itself holds: its theory chapter writes the equations of motion out (docs/theory.rst:16-24, one
    mode; the single-mode restatement at :49-52 drops the parentheses around the two terms, the code and :16-19 keep them;
    d_{n,lambda} = c_n r_{n,lambda}, so that the derivative of d w.r.t. R_n is the charge):
        nuclei   F_n,lambda = -( eps q_lambda + eps^2 / (m w^2) * sum_l d_l,lambda ) * c_n          lambda = x, y
        photon   m q''_lambda = -m w^2 q_lambda - eps * sum_n d_n,lambda
    evaluated here literally, in exact rational arithmetic, and compared with the oracle's forces."""
    from fractions import Fraction as Fr
    rng = np.random.default_rng(17)
    n = 40
    L = (30.0, 20.0, 25.0)
    pos = rng.uniform(-10, 10, (n, 3))
    tid = (np.arange(n) % 2).astype(np.int32)
    tid[-1] = 2
    charge = rng.uniform(-1, 1, n)
    charge[-1] = 0.0
    image = rng.integers(-2, 3, (n, 3)).astype(np.int32)
    omegac, eps, m = 0.31, 0.27, 1.7
    p = ref.make_params(omegac, eps, m)
    out = ref.compute(oracle_mod.pack_pos(pos, tid), charge, image, L, 2, p)
    r = [[Fr(float(pos[i, k])) + int(image[i, k]) * Fr(float(L[k])) for k in range(3)] for i in range(n)]
    q = r[-1]
    d = [sum(Fr(float(charge[i])) * r[i][k] for i in range(n - 1)) for k in range(3)]
    K = Fr(float(m)) * Fr(float(omegac)) ** 2                      # m w^2
    e = Fr(float(eps))
    for i in range(n - 1):
        for k in range(2):
            want = -(e * q[k] + e * e / K * d[k]) * Fr(float(charge[i]))
            assert abs(float(want) - out["force"][i, k]) <= 1e-13 * max(abs(float(want)), 1e-30), (i, k)
        assert out["force"][i, 2] == 0.0
    for k in range(2):
        want = -K * q[k] - e * d[k]
        assert abs(float(want) - out["force"][-1, k]) <= 1e-13 * abs(float(want))
    assert abs(float(-K * q[2]) - out["force"][-1, 2]) <= 1e-13 * abs(float(K * q[2]))   # z: free oscillator, no coupling


def test_no_photon_and_empty(ref, oracle_mod):
    pos4 = oracle_mod.pack_pos(np.zeros((3, 3)), np.array([0, 1, 0]))
    out = ref.compute(pos4, np.ones(3), np.zeros((3, 3), dtype=np.int32), (1, 1, 1), 2, ref.make_params(1, 1, 1))
    assert out["photon_idx"] == -1 and not out["force"].any() and not out["energies"].any()
    # L_typeid = -1 stands for "no type named L"
    out = ref.compute(pos4, np.ones(3), np.zeros((3, 3), dtype=np.int32), (1, 1, 1), -1, ref.make_params(1, 1, 1))
    assert out["photon_idx"] == -1
    empty = ref.compute(np.zeros((0, 4)), np.zeros(0), np.zeros((0, 3), dtype=np.int32), (1, 1, 1), 2,
                        ref.make_params(1, 1, 1))
    assert empty["photon_idx"] == -1 and empty["force"].shape == (0, 4)


def test_all_cores_courtesy_variant_agrees_to_rounding(ref, oracle_mod):
    """oracle/cavity_omp.c is only a timing courtesy for bench.py, but it must still compute the same physics."""
    rng = np.random.default_rng(8)
    n = 20_000
    L = (60.0, 60.0, 60.0)
    pos = rng.uniform(-30, 30, (n, 3))
    tid = (np.arange(n) % 2).astype(np.int32)
    tid[-1] = 2
    charge = rng.uniform(-1, 1, n)
    charge[-1] = 0.0
    image = rng.integers(-1, 2, (n, 3)).astype(np.int32)
    pos4 = oracle_mod.pack_pos(pos, tid)
    p = ref.make_params(0.0091, 1e-3, 1.0)
    a = ref.compute(pos4, charge, image, L, 2, p)
    b = oracle_mod.AllCoresCourtesy().compute(pos4, charge, image, L, 2, p)
    assert np.allclose(a["energies"], b["energies"], rtol=1e-10)
    assert np.abs(a["force"] - b["force"]).max() <= 1e-10 * np.abs(a["force"]).max()
