#!/usr/bin/env python3
"""Minimal use of the HIP cavity force outside HOOMD-blue (needs an MI355X; there is no CPU fallback).

The reference's driver does, inside a HOOMD simulation (examples/05_advanced_run.py:556-566 of cav-hoomd):

    cavityforce = hoomd.cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=g, omegac=omegac, phmass=1)
    integrator.forces.append(cavityforce)

Here the same object is attached to a stand-in for HOOMD's particle data that holds the arrays in GPU memory in
HOOMD's own layouts, evaluated a few times while the particles move, and polled the way the reference's trackers do.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "cav-hoomd_amd"))

import cavitymd  # noqa: E402
from cavitymd import PhysicalConstants, observables, synthetic, thermostats  # noqa: E402


def main():
    cfg = synthetic.config1(seed=1)              # 500 molecular particles + the photon ('L', appended last)
    pdata = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                              cfg["box"], device="cuda")
    omegac = PhysicalConstants.omegac_from_wavenumber(2000.0)
    force = cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=1e-3, omegac=omegac, phmass=1.0)
    force.attach(cavitymd.SystemDefinition(pdata))
    print("implementation:", force.implementation)

    field = observables.DensityField(pdata, observables.generate_fibonacci_sphere(50) * 1.0)
    n = pdata.getN()
    mass = np.where(cfg["typeid"] == 2, 1.0, 2.7e4)
    vel4 = torch.from_numpy(np.concatenate([np.zeros((n, 3)), mass[:, None]], axis=1)).cuda()   # HOOMD Scalar4 velocity

    for step in range(5):
        force.compute(step)                                   # one kernel launch, asynchronous
        print(f"step {step}: E_harmonic={force.harmonic_energy:.6e} E_coupling={force.coupling_energy:.6e} "
              f"E_dipole_self={force.dipole_self_energy:.6e} total={force.energy:.6e}")
        pdata.getPositions()[:, :3] += 1e-3 * torch.randn((n, 3), dtype=torch.float64, device="cuda")

    impl = force._force_impl
    print("total dipole:", observables.compute_total_dipole_moment(impl))
    print("cavity mode (KE, PE, total, T):", observables.cavity_mode(impl, vel4))
    S = observables.force_mass_sum(impl.workspace, impl.getForceArray(), vel4)
    print("sum |F|/m =", S, "-> dt =", observables.adaptive_timestep(1e-3, S))
    print("|rho(k)| for the first 3 wavevectors:", np.abs(field.compute()[:3]))
    print("forces on the first 2 particles:\n", force.forces[:2])

    # the Bussi reservoir thermostat of the reference's driver (examples/05_advanced_run.py:780-804), applied to the
    # molecular group; HOOMD's integration method would call it once per step and rescale the velocities itself
    vel4[:, :3] = 1e-3 * torch.randn((n, 3), dtype=torch.float64, device="cuda")
    bussi = thermostats.BussiReservoir(kT=PhysicalConstants.KB_HARTREE_PER_K * 100.0, tau=5.0)
    bussi.attach(n, members=np.arange(n - 1, dtype=np.uint32))
    rng = np.random.default_rng(0)
    for step in range(3):
        alpha, _ = bussi.step(step, 1.0, vel4, translational_dof=3.0 * (n - 1) - 3.0, rng=rng)
        print(f"thermostat step {step}: alpha={alpha:.6f} reservoir={bussi.total_reservoir_energy:.6e}")
    # the same step without a host round trip: kinetic energy, the rule and the rescaling all on the device, asynchronous;
    # the counters are fetched when they are read
    for step in range(3, 6):
        bussi.step_async(step, 1.0, vel4, translational_dof=3.0 * (n - 1) - 3.0, rng=rng)
    print(f"after three on-device steps: last alpha={bussi.device_state().last_alpha:.6f} "
          f"reservoir={bussi.total_reservoir_energy:.6e}")


if __name__ == "__main__":
    main()
