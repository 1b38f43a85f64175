"""CPU restatement of the reference's observables that sit next to the cavity-force path -- TEST INFRASTRUCTURE.

Follows src/cavitymd/analysis.py of muhammadhasyim/cav-hoomd (which cannot be imported here: it does `import hoomd`
at module top; the functions below are plain numpy and are restated one for one):

    compute_total_dipole_moment   analysis.py:18-31     np.dot(charge, unwrapped_positions), ALL particles
    compute_density_field         analysis.py:34-47     rho(k) = sum_j exp(i k.r_j) on the WRAPPED positions, all particles
    generate_fibonacci_sphere     analysis.py:50-64     the default wavevector directions (FieldAutocorrelationTracker,
                                                        analysis.py:296-306: kmag * 50 points)
    cavity mode properties        analysis.py:1324-1368 KE = 1/2 m v.v of the photon, T = (2/3) KE / k_B,
                                                        PE = the force's harmonic energy

    adaptive-dt reduction         simulation.py:66-92   sum_i |f_i| / m_i, dt = sqrt(tol / sum)

Pinning status: the reference has no tests for these either -> unpinned by reference fixtures; pinned here by closed-form
cases (tests/test_oracle_observables.py).
"""
from __future__ import annotations

import math

import numpy as np

KB_HARTREE_PER_K = 3.167e-6  # src/cavitymd/utils.py:13


def total_dipole_moment(position, image, charge, box_L) -> np.ndarray:
    unwrapped = np.asarray(position) + np.asarray(image) * np.asarray(box_L)[None, :]
    return np.dot(np.asarray(charge), unwrapped)


def fibonacci_sphere(samples: int = 100) -> np.ndarray:
    """Golden-angle spiral on the unit sphere, y from +1 to -1 (analysis.py:50-64)."""
    points = np.zeros((samples, 3))
    phi = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        radius = math.sqrt(1 - y * y)
        theta = phi * i
        points[i, 0] = math.cos(theta) * radius
        points[i, 1] = y
        points[i, 2] = math.sin(theta) * radius
    return points


def density_field(position, wavevectors) -> np.ndarray:
    """rho(k) for each wavevector, complex128; loop over wavevectors exactly as the reference does."""
    position = np.asarray(position, dtype=np.float64)
    re = np.zeros(len(wavevectors))
    im = np.zeros(len(wavevectors))
    for i, k in enumerate(np.asarray(wavevectors, dtype=np.float64)):
        kr = np.dot(position, k)
        re[i] = np.sum(np.cos(kr))
        im[i] = np.sum(np.sin(kr))
    return re + 1j * im


def density_field_exact(position, wavevectors) -> np.ndarray:
    """Same sums with exactly rounded accumulation (math.fsum) and k.r = (x kx + y ky) + z kz without FMA: the
    yardstick the GPU kernel is compared against (numpy's BLAS dot may fuse or reorder the three products)."""
    position = np.asarray(position, dtype=np.float64)
    out = np.zeros(len(wavevectors), dtype=np.complex128)
    for i, k in enumerate(np.asarray(wavevectors, dtype=np.float64)):
        kr = (position[:, 0] * k[0] + position[:, 1] * k[1]) + position[:, 2] * k[2]
        out[i] = math.fsum(np.cos(kr).tolist()) + 1j * math.fsum(np.sin(kr).tolist())
    return out


def cavity_mode(velocity, mass, typeid, harmonic_energy: float, L_typeid: int = 2):
    """(kinetic, potential, total, temperature) of the cavity oscillator; zeros if there is no photon
    (analysis.py:1329-1333).  The reference looks the photon up by `typeid == 2`; here the id is a parameter."""
    mask = np.asarray(typeid) == L_typeid
    if not np.any(mask):
        return 0.0, 0.0, 0.0, 0.0
    m = np.asarray(mass)[mask][0]
    v = np.asarray(velocity)[mask][0]
    ke = 0.5 * m * np.sum(v**2)
    total = ke + harmonic_energy
    return float(ke), float(harmonic_energy), float(total), float((2.0 / 3.0) * ke / KB_HARTREE_PER_K)


def force_mass_sum(total_forces, masses) -> float:
    """sum_i |f_i| / m_i exactly as AdaptiveTimestepUpdater.act spells it (simulation.py:84-86)."""
    total_forces = np.asarray(total_forces, dtype=np.float64)
    force_norm = np.array([np.linalg.norm(f) for f in total_forces])
    return float(np.sum(force_norm / np.asarray(masses, dtype=np.float64)))


def force_mass_sum_exact(total_forces, masses) -> float:
    f = np.asarray(total_forces, dtype=np.float64)
    n = np.sqrt((f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1]) + f[:, 2] * f[:, 2])
    return math.fsum((n / np.asarray(masses, dtype=np.float64)).tolist())
