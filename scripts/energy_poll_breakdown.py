"""Where the extra microseconds of 'energies polled after every evaluation' go.
usage: python scripts/energy_poll_breakdown.py [N=1000000]
Prints host-side and device-side pieces measured separately on the HBM-cold ring bench.py uses."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
import numpy as np, torch, cavitymd
from cavitymd import synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
frames = max(2, int(np.ceil(2 * 256 * 2**20 / (84 * (n + 1)))))
cfg = synthetic.diatomic_box(n, seed=1, finite_q=True)
ring = []
for f in range(min(frames, 64)):
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"], cfg["box"], device="cuda")
    p = cfg["params"]
    ring.append(cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"], p["phmass"]))
nf = len(ring)
for c in ring:
    c.compute(0)
    c.getEnergies()
torch.cuda.synchronize()
steps = 400

def loop(poll):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(steps):
        c = ring[s % nf]; c.compute(s)
        if poll: c.getEnergies()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

a = [loop(False) for _ in range(5)]; b = [loop(True) for _ in range(5)]
# host cost of the enqueue alone (queue kept shallow: sync every 8 calls so the queue never blocks the host)
t_enq = []
for rep in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(8): ring[s % nf].compute(s)
    t_enq.append((time.perf_counter() - t0) / 8 * 1e6)
    torch.cuda.synchronize()
# isolated evaluation from an idle GPU: enqueue -> energies visible on the host, and enqueue -> stream idle
t_flag, t_sync, t_get = [], [], []
for rep in range(200):
    c = ring[rep % nf]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.compute(rep); c.getEnergies(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    c.getEnergies(); t3 = time.perf_counter()
    t_flag.append((t1 - t0) * 1e6); t_sync.append((t2 - t0) * 1e6); t_get.append((t3 - t2) * 1e6)
# device time of the kernel (dispatch timestamps)
for c in ring: c.workspace.profile_enable(True)
for s in range(200): ring[s % nf].compute(s)
tot, cnt = 0.0, 0
for c in ring:
    ms, k = c.workspace.profile_read(); tot += sum(ms); cnt += k; c.workspace.profile_enable(False)
med = lambda v: float(np.median(v))
print(f"N={n+1} frames={nf}")
print(f"per evaluation, back to back, no polling      : {med(a):6.2f} us")
print(f"per evaluation, energies polled every step    : {med(b):6.2f} us   (+{med(b)-med(a):.2f})")
print(f"kernel(s) device time per evaluation           : {tot/cnt*1e3:6.2f} us")
print(f"host cost of one enqueue (compute() returns)   : {med(t_enq):6.2f} us")
print(f"isolated: enqueue -> energies on the host      : {med(t_flag):6.2f} us   (idle-queue launch latency + kernel up to the publishing block + PCIe write)")
print(f"isolated: enqueue -> stream idle (synchronize) : {med(t_sync):6.2f} us   (idle-queue launch latency + whole kernel + completion signal)")
print(f"getter on an already finished evaluation       : {med(t_get):6.2f} us   (flag already set: pure host call)")
print(f"=> with polling the host may not enqueue step k+1 before step k has published, so every launch is issued into an")
print(f"   (almost) empty queue: the launch latency that back-to-back enqueueing hides is exposed once per step.")
