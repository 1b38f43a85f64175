"""Interleaved A/B of one libcavmd tunable on whole evaluations (ring of frames, HBM-cold), several sizes.
usage: python scripts/ab_tunable.py NAME VALUE_A VALUE_B [N ...]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic
name, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sizes = [int(float(x)) for x in sys.argv[4:]] or [1_000_000]
for n in sizes:
    frames = max(2, int(np.ceil(2 * 256 * 2**20 / (84 * (n + 1)))))
    cfg = synthetic.diatomic_box(n, seed=1, finite_q=True)
    ring = []
    for f in range(min(frames, 64)):
        pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
        p = cfg["params"]
        ring.append(cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"]))
    steps = max(40, min(400, int(4e8 / n)))
    res = {va: [], vb: []}
    for rnd in range(7):
        for v in (va, vb):
            for c in ring: c.workspace.set_tunable(name, v)
            for s in range(len(ring)): ring[s % len(ring)].compute(s)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for s in range(steps): ring[s % len(ring)].compute(s)
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / steps * 1e6)
    print(f"N={n+1} frames={len(ring)} {name}: " + "  ".join(f"{v} -> median {np.median(res[v]):.2f} us (min {min(res[v]):.2f})" for v in (va, vb)))
    del ring; torch.cuda.empty_cache()
