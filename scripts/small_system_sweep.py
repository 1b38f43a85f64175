import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic
for n in (500, 1000, 2000, 4000, 8000, 16000, 32000):
    cfg = synthetic.diatomic_box(n, seed=1)
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
    p = cfg["params"]
    out = {}
    for mode, maxn in (("single-block", 1 << 20), ("two-launch", 0)):
        comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
        comp.workspace.set_tunable("small_system_max_n", maxn)
        for _ in range(50): comp.compute(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2000): comp.compute(0)
        torch.cuda.synchronize(); out[mode] = (time.perf_counter() - t0) / 2000 * 1e6
        out[mode + "_E"] = comp.getEnergies()
    print(n + 1, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items() if not k.endswith("_E")},
          "energies agree:", np.allclose(out["single-block_E"], out["two-launch_E"], rtol=1e-13))
