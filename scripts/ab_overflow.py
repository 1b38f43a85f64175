"""Single launch with an LDS budget (tiles beyond it re-read, pipelined) against two launches, large N.
usage: python scripts/ab_overflow.py N [N ...]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic
for n in [int(float(x)) for x in sys.argv[1:]]:
    frames = max(2, int(np.ceil(2 * 256 * 2**20 / (84 * (n + 1)))))
    cfg = synthetic.diatomic_box(n, seed=1, finite_q=True)
    ring = []
    for f in range(min(frames, 8)):
        pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
        p = cfg["params"]
        ring.append(cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"]))
    steps = max(20, min(400, int(4e8 / n)))
    variants = {"two launches": {"persistent": 0}, "single, 152 KB LDS": {"persistent": 1, "persistent_lds_kb": 0},
                "single, 76 KB LDS": {"persistent": 1, "persistent_lds_kb": 76}, "single, 32 KB LDS": {"persistent": 1, "persistent_lds_kb": 32}}
    res = {k: [] for k in variants}
    ref_force = None
    for rnd in range(5):
        for name, tun in variants.items():
            for c in ring:
                for k, v in tun.items(): c.workspace.set_tunable(k, v)
            for s in range(len(ring)): ring[s % len(ring)].compute(s)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for s in range(steps): ring[s % len(ring)].compute(s)
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / steps * 1e6)
            if rnd == 0:
                f0 = ring[0].getForceArray().clone()
                if ref_force is None: ref_force = f0
                else: assert torch.equal(f0, ref_force), name      # same partition, same fold: same bits
    print(f"N={n+1} frames={len(ring)}: " + "  ".join(f"{k} {np.median(v):.2f} us" for k, v in res.items()))
    del ring; torch.cuda.empty_cache()
