"""Pins the observable restatements in oracle/observables.py with closed-form cases.  No GPU."""
import math

import numpy as np

from cavitymd import observables as prod
from oracle import observables as obs


def test_fibonacci_sphere_matches_the_restated_reference_construction():
    for n in (2, 50, 100):
        a, b = obs.fibonacci_sphere(n), prod.generate_fibonacci_sphere(n)
        assert a.shape == (n, 3) and np.allclose(a, b, rtol=0, atol=1e-15)
        assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-12)
    pts = obs.fibonacci_sphere(50)
    assert pts[0, 1] == 1.0 and pts[-1, 1] == -1.0          # y runs from +1 to -1
    assert np.all(np.diff(pts[:, 1]) < 0)


def test_density_field_closed_forms():
    # one particle at the origin: rho = 1 for every k
    k = obs.fibonacci_sphere(7) * 2.5
    assert np.allclose(obs.density_field(np.zeros((1, 3)), k), 1.0 + 0j)
    # particles on a lattice commensurate with k: every phase is a multiple of 2 pi
    pos = np.array([[i, 0.0, 0.0] for i in range(8)])
    kk = np.array([[2 * math.pi, 0, 0], [math.pi, 0, 0], [math.pi / 2, 0, 0]])
    rho = obs.density_field(pos, kk)
    assert np.allclose(rho, [8.0, 0.0, 0.0], atol=1e-12)     # sum of (-1)^i = 0, sum of i^i over two periods = 0
    # rho(-k) = conj(rho(k)), |rho| <= N
    rng = np.random.default_rng(0)
    p = rng.uniform(-5, 5, (200, 3))
    a, b = obs.density_field(p, k), obs.density_field(p, -k)
    assert np.allclose(a, np.conj(b), atol=1e-12) and np.all(np.abs(a) <= 200)
    # the exactly-summed variant agrees with the plain one to rounding
    assert np.allclose(obs.density_field_exact(p, k), a, atol=1e-11)


def test_total_dipole_and_cavity_mode():
    pos = np.array([[1.0, 2, 3], [-3, 0.5, 1], [0.25, -0.5, 2]])
    img = np.array([[1, 0, 0], [0, 0, -1], [0, 0, 0]])
    d = obs.total_dipole_moment(pos, img, np.array([1.0, -0.5, 2.0]), (16.0, 16.0, 16.0))
    assert np.array_equal(d, [17 + 1.5 + 0.5, 2 - 0.25 - 1.0, 3 + 7.5 + 4.0])
    ke, pe, tot, T = obs.cavity_mode(np.array([[0, 0, 0], [1.0, 2.0, 2.0]]), np.array([1.0, 0.5]), np.array([0, 2]), 0.125)
    assert (ke, pe, tot) == (0.5 * 0.5 * 9.0, 0.125, 2.375) and T == (2.0 / 3.0) * 2.25 / 3.167e-6
    assert obs.cavity_mode(np.zeros((2, 3)), np.ones(2), np.array([0, 1]), 1.0) == (0.0, 0.0, 0.0, 0.0)


def test_force_mass_sum_closed_form():
    f = np.array([[3.0, 4.0, 0.0], [0.0, 0.0, 2.0], [1.0, 2.0, 2.0]])
    m = np.array([5.0, 4.0, 0.5])
    assert obs.force_mass_sum(f, m) == 1.0 + 0.5 + 6.0 == obs.force_mass_sum_exact(f, m)
    assert prod.adaptive_timestep(0.3, 7.5) == (0.3 / 7.5) ** 0.5 and prod.adaptive_timestep(0.3, 0.0) is None


# ---- against numbers the reference's own analysis.py / simulation.py computed (tests/golden/reference_python_golden.npz) ---
import os  # noqa: E402

import pytest  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    with np.load(os.path.join(golden_dir, "reference_python_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def test_fibonacci_sphere_against_the_executed_reference(gold):
    """Same construction, but math.cos/sin here vs numpy's in the reference: <= 1 ulp per coordinate; the product's
    vectorised version (the wavevector sets users exchange with the reference) likewise."""
    for n in (2, 10, 50, 100):
        want = gold[f"fibonacci/{n}"]
        assert np.abs(obs.fibonacci_sphere(n) - want).max() <= 2.3e-16
        assert np.abs(prod.generate_fibonacci_sphere(n) - want).max() <= 2.3e-16


def test_density_field_and_F_kt_against_the_executed_reference(gold):
    frames = gold["trajectory/frames"]
    n = frames.shape[1]
    for key in ("density/k1.0_n50", "density/k0.35_n17", "density/k2.5_n64"):
        k = gold[key + "/wavevectors"]
        nk = k.shape[0]
        kmag = float(key.split("/k")[1].split("_")[0])
        if nk == 50:                                                # the tracker's wavevectors = kmag * the sphere (:304-306)
            assert np.array_equal(k, gold["fibonacci/50"] * kmag)
        rho = np.array([obs.density_field(f, k) for f in frames])
        assert np.array_equal(rho, gold[key + "/rho_k"])            # the same numpy expressions: the same bits
        exact = np.array([obs.density_field_exact(f, k) for f in frames])
        assert np.abs(exact - gold[key + "/rho_k"]).max() <= 1e-13 * n
        # F(k,t) = mean_k Re(rho_k(0) conj(rho_k(t)))   (analysis.py:359-364)
        want = gold[key + "/F_kt"]
        got = [np.mean(np.real(rho[0] * np.conj(rho[t]))) for t in range(1, len(frames))]
        assert np.isnan(want[0]) and np.array_equal(got, want[1:])


def test_total_dipole_and_C_t_against_the_executed_reference(gold):
    frames = gold["trajectory/frames"]
    image, charge, box = (gold["force/n501_stand_in/" + k] for k in ("image", "charge", "box"))
    d = np.array([obs.total_dipole_moment(f, image, charge, box) for f in frames])
    assert np.array_equal(d, gold["dipole_acf/dipole_t"])
    assert np.array_equal([np.dot(d[0], x) for x in d], gold["dipole_acf/C_t"])


def test_cavity_mode_against_the_executed_reference(gold):
    for i in range(3):
        g = {k: gold[f"cavity_mode/{i}/{k}"] for k in ("typeid", "mass", "velocity", "harmonic_energy", "properties")}
        got = obs.cavity_mode(g["velocity"], g["mass"], g["typeid"], float(g["harmonic_energy"]), L_typeid=2)
        assert np.array_equal(got, g["properties"])
    assert not gold["cavity_mode/no_photon/properties"].any()


def test_adaptive_timestep_rule_against_the_executed_reference(gold):
    """AdaptiveTimestepUpdater.act: tol(t) = target - (target - initial) exp(-t/tau) with initial = 0.01 target, tau = 50 ps;
    S = sum |f_a + f_b| / m; dt = sqrt(tol / S); thermostat taus converted with ps_to_atomic_units."""
    from cavitymd.utils import PhysicalConstants as PC
    for i in range(3):
        g = {k: gold[f"adaptive_dt/{i}/{k}"] for k in ("mass", "force_a", "force_b", "elapsed_ps", "error_tolerance", "dt", "tau")}
        tol = 1e-3 - (1e-3 - 1e-3 * 0.01) * np.exp(-float(g["elapsed_ps"]) / 50.0)
        assert tol == float(g["error_tolerance"])
        S = obs.force_mass_sum(g["force_a"] + g["force_b"], g["mass"])
        assert np.sqrt(tol / S) == float(g["dt"])
        assert prod.adaptive_timestep(tol, S) == pytest.approx(float(g["dt"]), rel=4e-16)
        assert abs(obs.force_mass_sum_exact(g["force_a"] + g["force_b"], g["mass"]) - S) <= 1e-13 * S
        assert np.array_equal(g["tau"], [PC.ps_to_atomic_units(5.0), PC.ps_to_atomic_units(0.5)])
    assert float(gold["adaptive_dt/zero_force/dt"]) == 0.5      # S == 0: dt untouched (simulation.py:88)


def test_kinetic_energies_against_the_executed_reference(gold, oracle_mod):
    """EnergyTracker's internal kinetic energies (analysis.py:524-598): molecular = every particle whose typeid is not 2,
    cavity = the particle of typeid 2.  The oracle's kinetic energy (oracle/bussi_ref.c, the sum the thermostat consumes) over
    the same group: numpy sums pairwise, the oracle left to right -> 1e-13 relative; the exactly rounded sum likewise."""
    bussi = oracle_mod.BussiOracle()
    for i in range(3):
        g = {k: gold[f"kinetic/{i}/{k}"] for k in ("typeid", "mass", "velocity", "molecular", "cavity")}
        vel4 = np.concatenate([g["velocity"], g["mass"][:, None]], axis=1)
        members = np.flatnonzero(g["typeid"] != 2).astype(np.uint32)
        ke = bussi.kinetic_energy(vel4, members)
        assert abs(ke - g["molecular"][0]) <= 1e-13 * g["molecular"][0]
        hi, _ = bussi.kinetic_energy(vel4, members, exact=True)
        assert abs(hi - g["molecular"][0]) <= 1e-13 * g["molecular"][0]
        # the reference's temperature expression, reproduced to the bit from ITS kinetic energy
        assert (2.0 / 3.0) * g["molecular"][0] / (3 * len(members) * obs.KB_HARTREE_PER_K) == g["molecular"][1]
        cav = np.flatnonzero(g["typeid"] == 2)
        ke_c, _, _, _ = obs.cavity_mode(g["velocity"], g["mass"], g["typeid"], 0.0, L_typeid=2)
        assert len(cav) == 1 and ke_c == float(g["cavity"])
