"""Unit conversions and the unwrap helper exported by the package (names as in the reference's
``hoomd.cavitymd.utils``, src/cavitymd/utils.py:9-86, so user scripts keep working).

The numbers are physical constants / the reference's chosen conversion factors; tests pin them against
values produced by the reference's own module (tests/golden/utils_golden.json).
"""
from __future__ import annotations

import numpy as np


class PhysicalConstants:
    """Atomic-unit conversion factors used by cavity-MD scripts."""

    HARTREE_TO_CM_MINUS1 = 219474.63        # 1 Hartree in cm^-1
    KB_HARTREE_PER_K = 3.167e-6             # Boltzmann constant, Hartree / K
    ENERGY_JOULES = 4.35974e-18             # 1 Hartree in J
    LENGTH_METERS = 5.29177210544e-11       # 1 bohr in m
    MASS_KG = 9.1093837139e-31              # electron mass in kg
    TIME_SECONDS = 2.418884e-17             # atomic unit of time in s
    TIME_PS_CONVERSION = 2.418884e-5        # atomic unit of time in ps

    @classmethod
    def ps_to_atomic_units(cls, time_ps):
        return time_ps / cls.TIME_PS_CONVERSION

    @classmethod
    def atomic_units_to_ps(cls, time_au):
        return time_au * cls.TIME_PS_CONVERSION

    @classmethod
    def gamma_from_tau_ps(cls, tau_ps):
        """Langevin friction gamma = 1 / tau, with tau given in ps and gamma returned in atomic units."""
        if tau_ps <= 0.0:
            raise ValueError(f"tau_ps must be positive (got {tau_ps} ps): gamma = 1/tau is undefined otherwise")
        return 1.0 / cls.ps_to_atomic_units(tau_ps)

    @classmethod
    def omegac_from_wavenumber(cls, freq_cm_minus1):
        """Cavity frequency in atomic units from cm^-1 (examples/05_advanced_run.py:462, 562 of the reference)."""
        return freq_cm_minus1 / cls.HARTREE_TO_CM_MINUS1


def unwrap_positions(positions, images, box_lengths):
    """positions + images * box_lengths, row-wise: (N,3), (N,3), (3,) -> (N,3)."""
    p = np.asarray(positions)
    i = np.asarray(images)
    L = np.asarray(box_lengths)
    return p + i * L[None, :]
