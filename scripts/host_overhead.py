"""Host-side cost of one evaluation at the reference's production size (N = 501): Python wrapper vs raw ctypes vs pybind11."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import torch, cavitymd
from cavitymd import synthetic, _capi, _cavitymd
cfg = synthetic.config1(seed=1)
pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
p = cfg["params"]; n = pd.getN(); L = cfg["box"]
comp = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"])
force = comp.getForceArray()
M = 20000
def bench(name, fn):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(M): fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{name:28s} host enqueue {1e6*t_host/M:6.2f} us/eval   wall {1e6*t_all/M:6.2f} us/eval")
bench("CavityForceComputeHIP.compute", lambda: comp.compute(0))
ws = comp.workspace; prm = comp._params
args = (0, n, pd.getPositions().data_ptr(), pd.getCharges().data_ptr(), pd.getImages().data_ptr(), L, 2, prm, force.data_ptr())
bench("raw ctypes Workspace call", lambda: ws.compute_hoomd(*args))
ext = _cavitymd.CavityForceComputeHIP(n, p["omegac"], p["couplstr"], p["phmass"])
a = (pd.getPositions().data_ptr(), pd.getCharges().data_ptr(), pd.getImages().data_ptr(), n, L[0], L[1], L[2], 2, force.data_ptr(), 0)
bench("pybind11 computeForces", lambda: ext.computeForces(*a))
comp.workspace.profile_enable(True)
for _ in range(300): comp.compute(0)
ms, cnt = comp.workspace.profile_read()
print("GPU-side kernel time (dispatch timestamps): %.2f us per evaluation over %d" % (1e3 * ms[0] / cnt, cnt))
