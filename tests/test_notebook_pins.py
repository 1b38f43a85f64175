"""Pins taken from the reference's OWN recorded run: the output cells of examples/05_advanced_run.ipynb hold numbers the
reference printed while running on its init-0.gsd (tests/golden/notebook_recorded_outputs.json: data only).  They are
rounded as printed, so each check is "equal after rounding to the printed decimals" -- soft pins, but the only outputs of
the reference itself that exist for these formulas.  What they pin:

  * unit conversions and K = m omega_c^2                      (src/cavitymd/utils.py:12-21, src/CavityForceCompute.h:38-42)
  * the cavity-mode kinetic energy and its temperature          (src/cavitymd/analysis.py:1348-1366) -- row f2
  * the adaptive-timestep rule dt = sqrt(tol / sum |F_i|/m_i)   (src/cavitymd/simulation.py:66-92)   -- row f4
  * the F(k,t) reference interval in steps                      (src/cavitymd/analysis.py, time-based intervals)

The force path itself (dipole, forces, energies) has no recorded number in the reference: it stays unpinned.
"""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rec():
    with open(os.path.join(HERE, "golden", "notebook_recorded_outputs.json")) as f:
        return json.load(f)


def printed(x, decimals):
    return round(float(x), decimals)


def sig(x, digits):
    return float(f"{float(x):.{digits - 1}e}")


def test_unit_conversions_and_spring_constant(rec):
    from cavitymd import PhysicalConstants as PC
    from cavitymd import _capi
    assert printed(PC.ps_to_atomic_units(rec["timestep_ps"]), 6) == rec["timestep_au_printed"]
    assert printed(PC.ps_to_atomic_units(rec["tau_ps"]), 6) == rec["tau_au_printed"]
    assert printed(PC.KB_HARTREE_PER_K * rec["temperature_K"], 6) == rec["kT_au_printed"]
    assert printed(PC.gamma_from_tau_ps(rec["tau_ps"]), 6) == rec["langevin_base_gamma_printed"]
    omegac = PC.omegac_from_wavenumber(2000.0)
    assert sig(omegac, 6) == rec["omegac_printed"]
    # K as the C ABI computes it (host arithmetic: no GPU needed) and as the oracle does
    assert sig(_capi.make_params(omegac, rec["couplstr_printed"], 1.0).K, 6) == rec["K_printed"]
    import oracle
    assert sig(oracle.RefOracle().make_params(omegac, rec["couplstr_printed"], 1.0)["K"], 6) == rec["K_printed"]


def test_cavity_mode_kinetic_energy_and_temperature_oracle(rec):
    import oracle.observables as obs
    v = np.array([[0.0, 0.0, 0.0], rec["photon_initial_velocity"]])
    ke, pe, tot, T = obs.cavity_mode(v, np.array([1.0, rec["photon_mass"]]), np.array([0, 2]), 0.0)
    assert printed(ke, 6) == rec["photon_initial_KE_printed"]
    assert printed(T, 1) == rec["photon_expected_temperature_K_printed"]
    assert pe == 0.0 and tot == ke


def test_adaptive_timestep_rule(rec):
    from cavitymd import PhysicalConstants as PC
    from cavitymd.observables import adaptive_timestep
    dt = adaptive_timestep(rec["adaptive_error_tolerance"], rec["force_mass_sum_printed"])
    assert printed(dt, 5) == printed(rec["optimal_dt_au_printed"], 5)          # the sum itself is printed to 7 digits
    assert printed(PC.atomic_units_to_ps(dt) * 1000.0, 3) == rec["optimal_dt_fs_printed"]
    # F(k,t) reference interval: 1 ps at that timestep, in steps
    steps = int(PC.ps_to_atomic_units(rec["reference_interval_ps"]) / rec["optimal_dt_au_printed"])
    assert steps == rec["reference_interval_steps_printed"]


@pytest.mark.gpu
def test_cavity_mode_kernel_against_the_recorded_run(rec):
    """The same pin through the product: cavmd_cavity_mode on a two-particle system whose photon carries the recorded velocity."""
    import torch
    import cavitymd
    from cavitymd import observables as prod
    pos = np.array([[0.5, 0.25, -0.75], [0.0, 0.0, 0.0]])
    pd = cavitymd.ParticleData.from_arrays(pos, np.array([0, 2], dtype=np.int32), np.array([0.3, 0.0]),
                                           np.zeros((2, 3), dtype=np.int32), ["O", "N", "L"], (40.0, 40.0, 40.0), device="cuda")
    omegac = cavitymd.PhysicalConstants.omegac_from_wavenumber(2000.0)
    comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), omegac, rec["couplstr_printed"], 1.0)
    comp.compute(0)
    vel4 = torch.tensor([[0.0, 0.0, 0.0, 1.0], rec["photon_initial_velocity"] + [rec["photon_mass"]]], dtype=torch.float64,
                        device="cuda")
    ke, pe, tot, T = prod.cavity_mode(comp, vel4)
    assert printed(ke, 6) == rec["photon_initial_KE_printed"]
    assert printed(T, 1) == rec["photon_expected_temperature_K_printed"]
    assert pe == 0.0                                           # photon at the origin: no harmonic energy
