#!/usr/bin/env python3
"""Calls the kernels next to the force path (rows f2-f4) at N = 1e6 so that `rocprofv3 --kernel-trace --stats -- python3
scripts/profile_observables.py` prices them one by one; also prints the host-side wall time per call (enqueue to result on
the host for the calls that return a number, enqueue to drained stream for the others).

    kinetic energy (all particles / a 50 % index-list group), velocity rescale, sum |F|/m, rho(k) x 50, cavity mode,
    the fused on-device thermostat step
"""
import json
import os
import sys
import time

R = os.environ.get("GRAFT_REPO_ROOT", os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cavitymd  # noqa: E402
from cavitymd import _capi, observables, synthetic, thermostats  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = synthetic.diatomic_box(N, seed=1, finite_q=True, image_range=1)
pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
n = pd.getN()
rng = np.random.default_rng(0)
vel = np.empty((n, 4))
vel[:, :3] = rng.normal(0, 1e-3, (n, 3))
vel[:, 3] = rng.uniform(1.0, 30.0, n)
dvel = torch.from_numpy(vel).cuda()
frc = torch.from_numpy(np.concatenate([rng.normal(0, 1e-3, (n, 3)), np.zeros((n, 1))], axis=1)).cuda()
members = torch.from_numpy(np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint32)).cuda()
ws = _capi.Workspace(n)
out = {"N": n, "reps": REPS}


def wall(fn, reps=REPS, sync_each=False):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


out["kinetic_energy_all_us"] = wall(lambda: ws.kinetic_energy(0, dvel.data_ptr(), None, n))
out["kinetic_energy_group_half_us"] = wall(lambda: ws.kinetic_energy(0, dvel.data_ptr(), members.data_ptr(), n // 2))
out["scale_velocities_all_us_back_to_back"] = wall(lambda: ws.scale_velocities(0, dvel.data_ptr(), None, n, 1.0))
out["scale_velocities_group_half_us_back_to_back"] = wall(lambda: ws.scale_velocities(0, dvel.data_ptr(), members.data_ptr(), n // 2, 1.0))
out["force_mass_sum_us"] = wall(lambda: ws.force_mass_sum(0, n, frc.data_ptr(), dvel.data_ptr()))
field = observables.DensityField(pd, observables.generate_fibonacci_sphere(50))
out["density_field_50k_us_back_to_back"] = wall(field.enqueue, reps=max(REPS // 4, 10))
p = cfg["params"]
comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
comp.compute(0)
out["cavity_mode_us"] = wall(lambda: observables.cavity_mode(comp, dvel))
th = thermostats.BussiReservoir(kT=3.167e-4, tau=5.0)
th.attach(n, members=None, device="cuda")
dof = 3.0 * n - 3.0
vrng = np.random.default_rng(3)
variates = [thermostats.draw_variates(vrng, dof) for _ in range(64)]   # varying draws: alpha != 1, every rescale kernel does its work
count = [0]


def host_step():
    count[0] += 1
    th.step(count[0], 1.0, dvel, dof, variates=variates[count[0] % 64])


def device_step():
    count[0] += 1
    th.step_async(count[0], 1.0, dvel, dof, variates=variates[count[0] % 64])


out["bussi_step_host_rule_us"] = wall(host_step)
out["bussi_step_on_device_us_back_to_back"] = wall(device_step)
out["bussi_device_state"] = {k: getattr(th.device_state(), k) for k in ("steps", "refused", "last_alpha")}
print(json.dumps(out))
