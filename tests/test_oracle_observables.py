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
