"""The oracle against numbers the REFERENCE'S OWN PYTHON computed (tests/golden/reference_python_golden.npz, written by
tests/golden/make_reference_python_golden.py in the build container by executing src/cavitymd/cavity_force_python.py,
forces.py, analysis.py and simulation.py of the reference on an arithmetic-free `hoomd` stand-in).  No GPU.

What is pinned and how tightly
  * The reference holds two implementations of the force: the C++ class (restated in oracle/cavity_ref.c -- it cannot be
    built here) and the Python fallback (executed for these fixtures).  They are documented to coincide when there is exactly
    one cavity particle, it has typeid 1 and charge 0; on those cases the C oracle must reproduce the executed numbers.
  * The fallback sums the dipole with np.dot (BLAS order) and the C++ class left to right, so summed quantities agree at the
    rounding level only: |d_a - d_b| <= 2 N eps * sum|c_i r_i| (the a-priori bound for two summation orders), energies and
    forces follow with the pre-cancellation scales of the parity contract at 1e-12.
  * Everything downstream of the dipole is compared again with the dipole TAKEN from the fixture (formulas only, no
    summation-order slack): bit for bit wherever the reference's Python and the C++ class spell the expression the same way.
  * The two documented divergences (charged cavity particle; a second particle of the cavity type) must show up as
    divergences, and the restated mirror oracle/python_fallback_mirror.py must equal the executed fallback bit for bit
    everywhere (so the mirror, used by nothing else any more, is known to be a faithful copy).
"""
import os

import numpy as np
import pytest

from oracle import python_fallback_mirror as pyfb

EPS = np.finfo(float).eps
COINCIDE = ["n3", "n50_first", "n501_stand_in", "n2000_last", "n257_heavy_photon"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    with np.load(os.path.join(golden_dir, "reference_python_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def case(gold, name):
    pre = f"force/{name}/"
    return {k[len(pre):]: v for k, v in gold.items() if k.startswith(pre)}


def test_fixture_file_is_reference_executed(gold):
    assert "executing /root/reference/src/cavitymd" in str(gold["generated_by"])
    assert set(COINCIDE) < set(gold["force/names"].tolist())


@pytest.mark.parametrize("name", COINCIDE)
def test_c_oracle_reproduces_the_executed_python_force(ref, oracle_mod, gold, name):
    c = case(gold, name)
    omegac, g, m, K = c["params"]
    n = len(c["charge"])
    p = ref.make_params(omegac, g, m)
    assert p["K"] == K                                     # K = phmass * omegac**2, the same double
    a = ref.compute(oracle_mod.pack_pos(c["position"], c["typeid"]), c["charge"], c["image"], c["box"], 1, p)
    cav = int(np.flatnonzero(c["typeid"] == 1)[0])
    assert a["photon_idx"] == cav
    unwrapped = c["position"] + c["image"] * c["box"][None, :]
    dscale = np.abs(c["charge"][:, None] * unwrapped).sum(axis=0)
    assert np.all(np.abs(a["dipole"] - c["total_dipole"]) <= 2 * n * EPS * dscale + 1e-300)
    q = unwrapped[cav]
    dxy = np.abs(c["total_dipole"][:2]).max()
    # energies: E_h has no sum in it -> same expression, same bits; E_c, E_d carry the dipole's rounding
    assert a["energies"][0] == c["energies"][0]
    assert abs(a["energies"][1] - c["energies"][1]) <= 1e-12 * g * np.abs(q[:2]).max() * dxy
    assert abs(a["energies"][2] - c["energies"][2]) <= 1e-12 * c["energies"][2]
    S_mol = g * np.abs(c["charge"]) * (np.abs(q[:2]).max() + g / K * dxy)
    S_mol[cav] = K * np.abs(q).max() + g * dxy
    assert np.all(np.abs(a["force"][:, :3] - c["force"]) <= 1e-12 * S_mol[:, None])
    assert np.all(a["force"][:, 2][np.arange(n) != cav] == 0.0) and np.all(c["force"][:, 2][np.arange(n) != cav] == 0.0)
    assert np.all(a["force"][:, 3] == 0.0) and np.all(c["potential_energy"] == 0.0)


@pytest.mark.parametrize("name", COINCIDE)
def test_formulas_downstream_of_the_dipole_bit_for_bit(gold, name):
    """With d taken from the fixture, the C++ class's operator association (src/CavityForceCompute.cc:169-207, the one
    oracle/cavity_ref.c and the kernels use) is evaluated here in numpy scalars and compared with what the reference's Python
    produced from the same d.  Python spells F_i = -g * c_i * (q + (g/K) d) and the C++ class ((-g) c_i) * Dq: the same
    association; E_d = 0.5 * (g*g/K) * (d.d) in both; so equality is exact except where np.dot's two-term sums may fuse."""
    c = case(gold, name)
    omegac, g, m, K = c["params"]
    cav = int(np.flatnonzero(c["typeid"] == 1)[0])
    q = (c["position"] + c["image"] * c["box"][None, :])[cav]
    d = c["total_dipole"]
    Dq = np.array([q[0] + (g / K) * d[0], q[1] + (g / K) * d[1]])
    F = np.zeros((len(c["charge"]), 3))
    F[:, 0] = ((-g) * c["charge"]) * Dq[0]
    F[:, 1] = ((-g) * c["charge"]) * Dq[1]
    F[cav] = [-K * q[0] - g * d[0], -K * q[1] - g * d[1], -K * q[2]]
    assert np.array_equal(F, c["force"])
    E_h = 0.5 * K * (q[0] * q[0] + q[1] * q[1] + q[2] * q[2])
    E_c = g * (d[0] * q[0] + d[1] * q[1])
    E_d = 0.5 * (g * g / K) * (d[0] * d[0] + d[1] * d[1])
    assert np.allclose([E_h, E_c, E_d], c["energies"], rtol=4 * EPS, atol=0)
    assert c["total_cavity_energy"] == c["energies"][0] + c["energies"][1] + c["energies"][2]


def test_documented_divergences_are_real(ref, oracle_mod, gold):
    # 1. a charged cavity particle enters the fallback's dipole, not the C++ class's
    c = case(gold, "div_charged_cavity")
    omegac, g, m, K = c["params"]
    a = ref.compute(oracle_mod.pack_pos(c["position"], c["typeid"]), c["charge"], c["image"], c["box"], 1,
                    ref.make_params(omegac, g, m))
    cav = int(np.flatnonzero(c["typeid"] == 1)[0])
    r_cav = (c["position"] + c["image"] * c["box"][None, :])[cav]
    assert np.allclose(c["total_dipole"] - a["dipole"], c["charge"][cav] * r_cav, rtol=1e-9)
    assert not np.allclose(a["force"][:, :3], c["force"], rtol=1e-6)
    # 2. a second particle of the cavity type: force from the fallback, none from the C++ class (type test, .cc:191)
    c = case(gold, "div_two_cavity_typed")
    omegac, g, m, K = c["params"]
    a = ref.compute(oracle_mod.pack_pos(c["position"], c["typeid"]), c["charge"], c["image"], c["box"], 1,
                    ref.make_params(omegac, g, m))
    first, second = np.flatnonzero(c["typeid"] == 1)
    assert a["photon_idx"] == first
    assert np.all(a["force"][second] == 0.0) and np.any(c["force"][second] != 0.0)
    others = np.setdiff1d(np.arange(len(c["charge"])), [second])
    scale = np.abs(c["force"][others]).max()
    assert np.abs(a["force"][others, :3] - c["force"][others]).max() <= 1e-12 * scale
    # 3. nobody of the cavity type: zeros from both
    c = case(gold, "no_cavity")
    a = ref.compute(oracle_mod.pack_pos(c["position"], c["typeid"]), c["charge"], c["image"], c["box"], 1,
                    ref.make_params(*c["params"][:3]))
    assert a["photon_idx"] == -1 and not a["force"].any() and not a["energies"].any()
    assert not c["force"].any() and not c["energies"].any() and not c["potential_energy"].any()


def test_restated_mirror_equals_the_executed_fallback(gold):
    for name in gold["force/names"].tolist():
        c = case(gold, name)
        omegac, g, m, K = c["params"]
        b = pyfb.set_forces(c["position"], c["typeid"], c["image"], c["charge"], c["box"], g, omegac, m, cavity_typeid=1)
        assert np.array_equal(b["force"], c["force"]), name
        assert np.array_equal(b["energies"], c["energies"]), name
        if name != "no_cavity":
            assert np.array_equal(b["dipole"], c["total_dipole"]), name
