"""Time of one evaluation through cavmd_compute_soa: packed (N,3) arrays (generic strided kernels, three launches) vs HOOMD's
strided views of Scalar4 buffers (recognised and dispatched to the tuned AoS path)."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic, _capi
from cavitymd.state import type_tag_as_double
for n in (100_000, 1_000_000):
    cfg = synthetic.diatomic_box(n, seed=1, finite_q=True)
    N = n + 1
    frames = max(2, int(np.ceil(2 * 256 * 2**20 / (84 * N))))
    frames = min(frames, 48)
    prm = _capi.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], 1.0)
    ws = _capi.Workspace(N)
    packed, views = [], []
    for f in range(frames):
        pos3 = torch.from_numpy(cfg["position"]).cuda(); tid = torch.from_numpy(cfg["typeid"].astype(np.int32)).cuda()
        img = torch.from_numpy(cfg["image"]).cuda(); chg = torch.from_numpy(cfg["charge"]).cuda()
        frc3 = torch.empty((N, 3), dtype=torch.float64, device="cuda"); pe = torch.empty((N,), dtype=torch.float64, device="cuda")
        packed.append(((pos3.data_ptr(), 24), (tid.data_ptr(), 4), (img.data_ptr(), 12), (chg.data_ptr(), 8), (frc3.data_ptr(), 24), (pe.data_ptr(), 8), (pos3, tid, img, chg, frc3, pe)))
        pos4 = torch.from_numpy(np.concatenate([cfg["position"], type_tag_as_double(cfg["typeid"])[:, None]], axis=1)).cuda()
        frc4 = torch.empty((N, 4), dtype=torch.float64, device="cuda")
        views.append(((pos4.data_ptr(), 32), (pos4.data_ptr() + 24, 32), (img.data_ptr(), 12), (chg.data_ptr(), 8), (frc4.data_ptr(), 32), (frc4.data_ptr() + 24, 32), (pos4, frc4)))
    for name, ring in (("packed (N,3) arrays", packed), ("HOOMD Scalar4 views", views)):
        def step(s):
            a = ring[s % len(ring)]
            ws.compute_soa(0, N, a[0], a[1], a[2], a[3], cfg["box"], 2, prm, a[4], a[5])
        for s in range(20): step(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 300
        for s in range(steps): step(s)
        torch.cuda.synchronize()
        print(f"N={N} {name}: {(time.perf_counter() - t0) / steps * 1e6:.2f} us per evaluation")
