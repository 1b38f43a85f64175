"""Density field: lane = particle against lane = wavevector.  usage: python scripts/ab_density.py [N] [n_k ...]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic, observables
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
cfg = synthetic.diatomic_box(n, seed=1, finite_q=True)
pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
for n_k in [int(x) for x in sys.argv[2:]] or [50, 64, 100, 17]:
    field = observables.DensityField(pd, observables.generate_fibonacci_sphere(n_k) * 1.0)
    out = {}
    for mode in (0, 1, 2, 3, 0, 1, 2, 3):
        field._ws.set_tunable("rho_lane_particle", mode)
        for _ in range(3): field.enqueue()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): field.enqueue()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e6
        out.setdefault(mode, []).append(t)
        res = field.compute()
        out.setdefault(("r", mode), res)
    d = np.abs(out[("r", 0)] - out[("r", 1)]).max()
    print(f"N={n+1} n_k={n_k}: lane=wavevector {min(out[0]):.1f} us   lane=particle KC25 {min(out[1]):.1f} KC10 {min(out[2]):.1f} KC5 {min(out[3]):.1f} us   max |difference| {d:.2e}")
