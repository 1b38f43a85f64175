"""The reference contains TWO implementations of the cavity force (C++ class and Python fallback).  Our oracle follows
the C++ one; here it is checked against a restatement of the Python one on inputs where the two are documented to
coincide (one cavity particle, charge 0).  Agreement of two independently written reference sources is the closest
thing to a pin the reference offers for the force formulas.  No GPU."""
import numpy as np
import pytest

from oracle import python_fallback_mirror as pyfb


@pytest.mark.parametrize("seed,n", [(0, 3), (1, 50), (2, 501), (3, 2000)])
def test_cpp_restatement_agrees_with_python_fallback_restatement(ref, oracle_mod, seed, n):
    rng = np.random.default_rng(seed)
    L = (40.0, 37.5, 43.25)
    pos = rng.uniform(-0.5, 0.5, (n, 3)) * np.asarray(L)
    charge = rng.uniform(-1, 1, n)
    image = rng.integers(-2, 3, (n, 3)).astype(np.int32)
    tid = np.zeros(n, dtype=np.int32)           # molecules: type 0; the cavity particle: type 1 (the fallback's id)
    cav = int(rng.integers(0, n))
    tid[cav] = 1
    charge[cav] = 0.0
    g, omegac, m = 1e-3, 2000.0 / 219474.63, 1.0
    a = ref.compute(oracle_mod.pack_pos(pos, tid), charge, image, L, 1, ref.make_params(omegac, g, m))
    b = pyfb.set_forces(pos, tid, image, charge, L, g, omegac, m, cavity_typeid=1)
    assert a["photon_idx"] == b["cavity_idx"] == cav
    # numpy's np.dot may associate/fuse differently from the sequential C loop: compare at the rounding level
    dscale = np.abs(charge[:, None] * (pos + image * np.asarray(L))).sum(axis=0)
    assert np.all(np.abs(a["dipole"] - b["dipole"]) <= 4 * n * np.finfo(float).eps * dscale / n + 1e-300)
    assert np.allclose(a["energies"], b["energies"], rtol=1e-11, atol=0)
    q = pos[cav] + image[cav] * np.asarray(L)
    K = m * omegac * omegac
    scale = g * (np.abs(q[:2]).max() + g / K * np.abs(a["dipole"][:2]).max())
    assert np.abs(a["force"][:, :3] - b["force"]).max() <= 1e-11 * max(scale, K * np.abs(q).max())
    assert np.all(a["force"][:, 3] == 0.0)


def test_documented_divergences_are_real(ref, oracle_mod):
    """A charged cavity particle enters the Python fallback's dipole but not the C++ class's: the restatements must
    DISAGREE there, i.e. they really model two different reference code paths."""
    pos = np.array([[1.0, 2, 3], [0.25, -0.5, 2]])
    tid = np.array([0, 1], dtype=np.int32)
    image = np.zeros((2, 3), dtype=np.int32)
    charge = np.array([1.0, 4.0])
    a = ref.compute(oracle_mod.pack_pos(pos, tid), charge, image, (16, 16, 16), 1, ref.make_params(2.0, 0.5, 0.25))
    b = pyfb.set_forces(pos, tid, image, charge, (16, 16, 16), 0.5, 2.0, 0.25, cavity_typeid=1)
    assert np.array_equal(a["dipole"], [1.0, 2.0, 3.0]) and np.array_equal(b["dipole"], [2.0, 0.0, 11.0])
    assert not np.allclose(a["force"][:, :3], b["force"])
