#!/usr/bin/env python3
"""bench.py -- cavity-force evaluations per second on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config 3 / 5 -- 1e6 molecular particles in neutral diatomics + the
photon, finite-q start, g = 1e-3, omega_c = 2000 cm^-1 -- one independent replica per GPU (replica r has
seed r + 1).  A "step" is one evaluation of the cavity force (one launch at this size, cavmd_persistent_kernel.hpp) through the C ABI.  Successive
steps walk a ring of `frames` trajectory frames (positions perturbed as the thermostatted integrator would),
each with its own pos/charge/image/force arrays, sized so that the ring exceeds the 256 MiB Infinity Cache:
every step streams its 92 N algorithmic bytes from HBM, as it would inside a real MD step whose other kernels
have flushed the caches.  (The cache-hot figure, same frame every step, is reported under "extras".)

All inputs are resident in HBM before the timed region.  Timing: W warm-up steps, barrier + synchronize,
K steps, synchronize + barrier; the slowest rank's time is used; value = N_gpus * K / t.

One JSON line on rank 0, with
  roofline      dominant kernel's algorithmic bytes per launch / its mean launch duration (HIP events on the
                launch stream, taken in a second pass of K steps so the events do not perturb `value`),
                against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference's CavityForceCompute::computeForces, one thread) timed
                on this box's host cores on the same workload for a bounded number of evaluations
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "cav-hoomd_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def launch_ranks(n_ranks, argv, script=None):
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh child processes (one per LOCAL_RANK, the
    environment torch.distributed.run would give them), relay their output, print rank 0's JSON line LAST and return
    the worst exit code.  (`script`: the program the ranks run, this file unless a test substitutes a probe.)  Called before torch / the HIP library are imported: this parent never touches a GPU and
    nothing is exec'ed after GPU initialisation (the reference runs its replicas as separate processes too:
    one SLURM array task per replica, submit.sh:3)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    out0_file = tempfile.TemporaryFile(mode="w+")   # rank 0's stdout: holds the JSON line, relayed when all ranks are done
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0_file if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    # Wait for all ranks; if one fails, the others would sit in a collective until its time-out: end them (exactly the
    # children started above) after a short grace period instead.
    codes = [None] * n_ranks
    failed_at = None
    while any(c is None for c in codes):
        for k, pr in enumerate(procs):
            if codes[k] is None:
                codes[k] = pr.poll()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 20.0:
            for k, pr in enumerate(procs):
                if codes[k] is None:
                    pr.kill()
                    codes[k] = pr.wait()
        time.sleep(0.05)
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    lines = out0.splitlines()
    json_at = max((k for k, ln in enumerate(lines) if ln.startswith("{") and '"metric"' in ln), default=None)
    for k, ln in enumerate(lines):
        if k != json_at:
            print(ln, file=sys.stderr)
    if json_at is not None:
        sys.stdout.flush()
        print(lines[json_at], flush=True)
    worst = max((abs(c) for c in codes), default=0)
    if worst == 0 and json_at is None:
        worst = 1
    return worst


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    # decide before the heavy imports: the parent of a self-launched job must stay GPU-free
    _pre = argparse.ArgumentParser(add_help=False)
    _pre.add_argument("--gpus", type=int, default=1)
    _n = _pre.parse_known_args()[0].gpus
    if _n > 1:
        sys.exit(launch_ranks(_n, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cavitymd  # noqa: E402
from cavitymd import replicas, synthetic  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable on copies)
MALL_BYTES = 256 * 2**20
BYTES_REDUCE = 52             # pos 32 + charge 8 + image 12   (dipole_partials_kernel)
BYTES_MAP = 40                # charge 8 + force 32            (force_map_aos_fused_kernel)
BYTES_EVAL = BYTES_REDUCE + BYTES_MAP   # 92: also what the single-launch kernel is priced at (it moves 84)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n-molecular", type=int, default=1_000_000)
    ap.add_argument("--frames", type=int, default=0, help="trajectory frames in the ring (0 = enough to exceed 2x the Infinity Cache)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget for the cpu_baseline leg (split over the -O2 and -O3 builds)")
    ap.add_argument("--no-extras", action="store_true", help="skip the 1e5 / 1e7 / cache-hot side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class Frame:
    """One trajectory frame resident on the GPU with its own compute object (force array + workspace)."""

    def __init__(self, cfg, device):
        pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                               cfg["box"], device=device)
        p = cfg["params"]
        self.compute = cavitymd.CavityForceComputeHIP(cavitymd.SystemDefinition(pd), p["omegac"], p["couplstr"],
                                                      p["phmass"])
        self.n = pd.getN()


def build_ring(cfg, frames, device):
    ring = [Frame(cfg, device)]
    cur = cfg
    for k in range(1, frames):
        cur = synthetic.perturb(cur, k)
        ring.append(Frame(cur, device))
    return ring


def run_steps(ring, steps, first=0):
    nf = len(ring)
    for s in range(first, first + steps):
        ring[s % nf].compute.compute(s)


def prime(ring):
    """One untimed evaluation per frame, separate from --warmup: every frame's arrays, workspace and code path have been
    touched once before anything is timed (with --warmup 5 and a 7-frame ring, frames 5 and 6 used to run for the first
    time inside the timed region)."""
    run_steps(ring, len(ring))
    torch.cuda.synchronize()


def timed(ring, steps, warmup, ctx=None):
    prime(ring)
    run_steps(ring, warmup)
    if ctx is not None:
        replicas.barrier(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(ring, steps, first=warmup)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this rank's K steps on its own clock
    if ctx is not None:
        replicas.barrier(ctx)
    return time.perf_counter() - t0, own    # (the contract's bracket: barrier + synchronize on both sides, this rank's own)


def kernel_times(ring, steps, warmup):
    """Mean device time per launch of each kernel (dispatch timestamps via hipExtLaunchKernelGGL start/stop events on the
    launch stream) and the per-evaluation samples behind the means."""
    for f in ring:
        f.compute.workspace.profile_enable(True)
    run_steps(ring, max(warmup, len(ring)))
    for f in ring:
        f.compute.workspace.profile_read()  # discard warm-up
    run_steps(ring, steps, first=warmup)
    tot = [0.0, 0.0, 0.0]
    launches = 0
    samples = []
    for f in ring:
        samples.append(f.compute.workspace.profile_samples())
        ms, n = f.compute.workspace.profile_read()
        tot = [a + b for a, b in zip(tot, ms)]
        launches += n
        f.compute.workspace.profile_enable(False)
    samples = np.concatenate(samples, axis=0) if samples else np.zeros((0, 3))
    return [t / max(launches, 1) for t in tot], launches, samples


def percentiles(samples_ms):
    """median and p10/p90 (microseconds) of the per-evaluation kernel times, per kernel and for their sum."""
    if len(samples_ms) == 0:
        return None
    us = 1e3 * np.asarray(samples_ms)
    cols = {"reduce": us[:, 0], "map": us[:, 2], "evaluation_kernels": us.sum(axis=1)}
    return {k: {"p10": float(np.percentile(v, 10)), "median": float(np.percentile(v, 50)),
                "p90": float(np.percentile(v, 90))} for k, v in cols.items()}


def roofline_block(n, kt_ms):
    """kt_ms = mean device time of {slot 0, slot 1, slot 2}.  Single-launch evaluation (the default for 1024 < N <~ 2.4e6):
    slot 0 holds the one kernel and its algorithmic bytes are the evaluation's 92 N (a fused kernel that
    keeps the charges on-chip moves 84 N, the figure is still priced at 92 N).  Two launches: slot 0 = reduction (52 N),
    slot 2 = fused force map (40 N); three launches add the finalize kernel in slot 1 (no algorithmic bytes)."""
    if kt_ms[2] <= 0 and kt_ms[1] <= 0:
        names = ("cavity_persistent_kernel" if n > 1024 else "cavity_small_system_kernel", "", "")
        bytes_per_launch = (BYTES_EVAL * n, 0, 0)
    else:
        names = ("dipole_partials_kernel", "finalize_kernel", "force_map_aos_fused_kernel")
        bytes_per_launch = (BYTES_REDUCE * n, 0, BYTES_MAP * n)
    kernels = {}
    for name, b, t in zip(names, bytes_per_launch, kt_ms):
        if t <= 0:
            continue
        kernels[name] = {"avg_ms": t, "algorithmic_bytes": b, "GBps": (b / (t * 1e-3) / 1e9) if b else None}
    dom = 0 if kt_ms[0] >= kt_ms[2] else 2
    achieved = bytes_per_launch[dom] / (kt_ms[dom] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": bytes_per_launch[dom],
            "avg_launch_ms": kt_ms[dom], "kernels": kernels, "launches_per_evaluation": sum(1 for t in kt_ms if t > 0),
            "evaluation_GBps": BYTES_EVAL * n / (sum(kt_ms) * 1e-3) / 1e9}


def load_pmc_traffic(workload_n):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), if they were taken
    on this workload; otherwise None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        ent = d.get(str(workload_n))
        return ent
    except (OSError, ValueError):
        return None


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_s):
    """The oracle (port of the reference CPU path) on one host thread, bounded sample of the same workload."""
    import oracle
    out = {}
    pos4 = oracle.pack_pos(cfg["position"], cfg["typeid"])
    for opt in ("O2", "O3"):
        ref = oracle.RefOracle(opt)
        p = ref.make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
        args = (pos4, cfg["charge"], cfg["image"], cfg["box"], cfg["L_typeid"], p)
        t1 = ref.time_evaluations(*args, 2) / 2          # warm-up + estimate
        iters = int(max(3, min(5000, (budget_s / 2) / max(t1, 1e-9))))
        t = ref.time_evaluations(*args, iters)
        out[opt] = {"evals_per_s": iters / t, "iters": iters, "seconds": t}
    best = "O2"
    n = len(cfg["charge"])
    # courtesy: the same passes on every core of this host (OpenMP; NOT the reference's algorithm -- it is single-threaded)
    try:
        omp = oracle.AllCoresCourtesy()
        p = oracle.RefOracle("O2").make_params(cfg["params"]["omegac"], cfg["params"]["couplstr"], cfg["params"]["phmass"])
        args = (pos4, cfg["charge"], cfg["image"], cfg["box"], cfg["L_typeid"], p)
        t1 = omp.time_evaluations(*args, 5) / 5
        iters = int(max(5, min(5000, 3.0 / max(t1, 1e-9))))
        t = omp.time_evaluations(*args, iters)
        courtesy = {"evals_per_s": iters / t, "threads": omp.threads(), "iters": iters,
                    "note": "OpenMP variant, parallel reduction order: not the reference's algorithm, not an oracle"}
    except OSError as e:  # no libgomp on this host
        courtesy = {"error": str(e)}
    return {"value": out[best]["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{out[best]['iters']} evaluations of frame 0 of the same workload (N={n}), oracle/cavity_ref.c "
                      f"gcc -O2 -ffp-contract=off, 1 thread, {out[best]['seconds']:.1f} s",
            "GBps_equiv": BYTES_EVAL * n * out[best]["evals_per_s"] / 1e9,
            "O3_evals_per_s": out["O3"]["evals_per_s"], "host_cpus": os.cpu_count(), "host_cpu_model": host_cpu_model(),
            "all_cores_courtesy": courtesy}


def side_measurement(cfg, device, frames, steps, warmup):
    ring = build_ring(cfg, frames, device)
    t, _ = timed(ring, steps, warmup)
    kt, _, _ = kernel_times(ring, steps, warmup)
    n = ring[0].n
    launches = 1 if kt[2] == 0 else (2 if kt[1] == 0 else 3)
    out = {"N": n, "frames": frames, "evals_per_s": steps / t, "us_per_eval": 1e6 * t / steps,
           "evaluation_GBps_wall": BYTES_EVAL * n * steps / t / 1e9, "kernel_avg_us": [1e3 * x for x in kt],
           "evaluation_GBps_kernels": BYTES_EVAL * n / (sum(kt) * 1e-3) / 1e9 if sum(kt) > 0 else None,
           # two launches: slot 0 = reduction (52 N), slot 2 = fused force map (40 N); one launch: slot 0 holds all of it
           "reduce_GBps": BYTES_REDUCE * n / (kt[0] * 1e-3) / 1e9 if launches > 1 else None,
           "map_GBps": BYTES_MAP * n / (kt[2] * 1e-3) / 1e9 if launches > 1 else None,
           "launches_per_eval": launches}
    del ring
    torch.cuda.empty_cache()
    return out


def energy_poll_measurement(cfg, device, frames, steps, warmup):
    """Variant (ii) of SURVEY.md 8(d): every evaluation is followed by the three energy getters, as the reference's
    EnergyTracker does at period 1.  (The getters spin on a host-visible flag the publishing block sets; no memcpy.)"""
    ring = build_ring(cfg, frames, device)
    nf = len(ring)
    for s in range(warmup):
        ring[s % nf].compute.compute(s)
        ring[s % nf].compute.getEnergies()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        c = ring[s % nf].compute
        c.compute(s)
        c.getEnergies()
    t = time.perf_counter() - t0
    return {"N": ring[0].n, "frames": nf, "evals_per_s": steps / t, "us_per_eval": 1e6 * t / steps,
            "note": "evaluation + energy getters every step (host-visible result block, no memcpy, no stream sync)"}


def two_replicas_measurement(cfg, device, frames, steps, warmup):
    """NOT the headline: two independent replicas of the workload on ONE GPU, one stream each, enqueued alternately by the
    one host thread.  Each replica's evaluations stay strictly sequential (as in MD); the other replica's kernels fill the
    gaps the in-launch hand-off and the launch boundaries leave.  Two single-launch grids fit side by side."""
    rings = [build_ring(cfg, frames, device), build_ring(synthetic.perturb(cfg, 991), frames, device)]
    streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
    nf = frames

    def run(count, first=0):
        for s in range(first, first + count):
            for ring, st in zip(rings, streams):
                ring[s % nf].compute.compute(s, stream=st)

    run(nf)
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, first=warmup)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    return {"N": rings[0][0].n, "replicas_on_the_gpu": 2, "evals_per_s_total": 2 * steps / t,
            "us_per_eval_per_replica": 1e6 * t / steps,
            "note": "side measurement; the headline value is ONE replica per GPU, as BASELINE.json's north_star prescribes"}


def density_field_measurement(cfg, device, n_k=50, kmag=1.0, steps=20, warmup=3):
    """Row f3 side measurement: rho(k) for 50 Fibonacci-sphere wavevectors (the reference tracker's default)."""
    from cavitymd import observables
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device=device)
    field = observables.DensityField(pd, observables.generate_fibonacci_sphere(n_k) * kmag)
    for _ in range(warmup):
        field.enqueue()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        field.enqueue()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    n = pd.getN()
    # priced against fp64 VALU issue: 34 fp64 instructions per term (5 for k.r, 27 for the reduced sincos, 2 accumulations), one
    # wave instruction = 64 lanes in 4 cycles per SIMD -> 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3e12 lane-instructions/s;
    # with 50 of 64 lanes carrying a wavevector the reachable rate is 50/64 of that
    lane_rate = 256 * 4 * 16 * 2.4e9
    return {"N": n, "n_k": n_k, "us_per_call": 1e6 * t, "sincos_per_s": n * n_k / t,
            "fp64_instruction_issue_fraction": 34.0 * n * n_k / t / lane_rate,
            "fraction_of_reachable_with_idle_lanes": 34.0 * n * n_k / t / (lane_rate * n_k / (64.0 * math.ceil(n_k / 64))),
            "note": "fp64 instruction-issue bound (transcendentals by polynomial): N*n_k sincos per call, 34 fp64 instructions "
                    "each; positions are only 32 N bytes"}


def thermostat_measurement(n, device, steps=50, warmup=5):
    """Row f4 side measurement: one Bussi reservoir step = group kinetic energy on the GPU (32 N bytes read, result to the
    host: the scalar rule needs it), host arithmetic, velocity rescale (64 N bytes)."""
    from cavitymd import thermostats
    rng = np.random.default_rng(0)
    vel = np.empty((n, 4))
    vel[:, :3] = rng.normal(0, 1e-3, (n, 3))
    vel[:, 3] = 2.7e4
    dvel = torch.from_numpy(vel).to(device)
    th = thermostats.BussiReservoir(kT=3.167e-4, tau=5.0)
    th.attach(n, members=None, device=device)
    dof = 3.0 * n - 3.0
    var = [0.3, (dof - 1) / 2, 0.0, 0.0]
    for _ in range(warmup):
        th.step(0, 1.0, dvel, dof, variates=var)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        th.kinetic_energy(dvel)
    t_ke = (time.perf_counter() - t0) / steps
    vrng = np.random.default_rng(3)
    draws = [thermostats.draw_variates(vrng, dof) for _ in range(64)]   # varying draws: alpha != 1 in every step
    t0 = time.perf_counter()
    for s in range(steps):
        th.step(s, 1.0, dvel, dof, variates=draws[s % 64])
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    for s in range(warmup):
        th.step_async(s, 1.0, dvel, dof, variates=draws[s % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        th.step_async(s, 1.0, dvel, dof, variates=draws[s % 64])
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / steps
    return {"N": n, "kinetic_energy_us": 1e6 * t_ke, "kinetic_energy_GBps": 32 * n / t_ke / 1e9, "full_step_us": 1e6 * t_step,
            "full_step_on_device_us": 1e6 * t_dev, "full_step_on_device_GBps": 96 * n / t_dev / 1e9,
            "note": "host rule: KE = one kernel, the number reaches the host through a mapped flag, host arithmetic, then the "
                    "rescale kernel.  on device (cavmd_bussi_step_device): one partial per block, then every block of the rescale "
                    "kernel folds them, evaluates the rule and rescales; back to back, no host round trip; 96 N bytes per step "
                    "(32 N read + 64 N read/write).  Kernel times: profiles/r03/observables_kernel_stats.csv"}


def main():
    args = parse_args()
    ctx = replicas.init_from_env(prefer_gpu=True)
    if ctx.device.type != "cuda":
        raise SystemExit("bench.py needs a GPU: the cavity force has no CPU path in this package")
    if ctx.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world_size}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    frames = args.frames
    if frames <= 0:
        frames = max(2, math.ceil(2 * MALL_BYTES / (84 * (args.n_molecular + 1))))
    p = synthetic.default_params()
    spec = replicas.broadcast_spec(ctx, {
        "omegac": p["omegac"], "couplstr": p["couplstr"], "phmass": p["phmass"], "n_molecular": args.n_molecular,
        "base_seed": 0, "steps": args.steps, "warmup": args.warmup, "frames": frames, "finite_q": True,
    } if ctx.rank == 0 else None)

    # replica `rank + 1` of config 5 on this rank's GPU (seed = replica id)
    replica_id = ctx.rank + 1
    params = {"omegac": spec["omegac"], "couplstr": spec["couplstr"], "phmass": spec["phmass"]}
    cfg = synthetic.diatomic_box(spec["n_molecular"], seed=replicas.replica_seed(replica_id, spec["base_seed"]),
                                 finite_q=spec["finite_q"], image_range=1, params=params,
                                 name=f"config5_replica{replica_id}")
    ring = build_ring(cfg, spec["frames"], ctx.device)
    n = ring[0].n
    torch.cuda.synchronize()

    elapsed, own = timed(ring, spec["steps"], spec["warmup"], ctx)
    elapsed = replicas.max_over_ranks(ctx, elapsed)
    value = ctx.world_size * spec["steps"] / elapsed
    # every rank's own rate (its K steps over its own clock between the two barriers): an imbalanced node shows here
    per_rank = {"min": replicas.min_over_ranks(ctx, spec["steps"] / own), "max": replicas.max_over_ranks(ctx, spec["steps"] / own)}

    kt, launches, samples = kernel_times(ring, spec["steps"], spec["warmup"])
    roof = roofline_block(n, kt)
    roof["kernel_time_us_percentiles"] = percentiles(samples)
    pmc = load_pmc_traffic(n)
    if pmc is not None:
        roof["traffic"] = pmc.get(roof["kernel"], {}).get("hbm_bytes_per_launch")
        roof["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "workload taken by the builder and committed; NOT measured in this run")
        roof["traffic_detail"] = pmc

    line = {
        "metric": "cavity_force_evals_per_sec", "value": value, "unit": "evals/s", "n_gpus": ctx.world_size,
        "steps": spec["steps"], "warmup": spec["warmup"], "ms_per_step": 1e3 * elapsed / spec["steps"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config3/config5: 1e6 diatomic particles + photon, finite-q start, g=1e-3, "
                               "omegac=2000cm^-1; one independent replica per GPU (seed = rank+1); "
                               f"ring of {spec['frames']} trajectory frames (HBM-cold)",
                   "N_particles": n, "frames": spec["frames"], "primed_frames": spec["frames"],
                   "replicas": ctx.world_size,
                   "algorithmic_bytes_per_eval": BYTES_EVAL * n, "layout": "HOOMD AoS (Scalar4 pos/force, int3 image)",
                   "collectives_on_data_path": 0,
                   # the job's control collectives (start-up broadcast, barriers, MAX/MIN of the timing): "nccl" = RCCL
                   "dist_backend": ctx.backend, "dist_collective_device": str(ctx.coll_device) if ctx.backend else None},
        "per_rank_evals_per_s": per_rank,
        "achieved_GBps_wall": BYTES_EVAL * n * value / 1e9,
        "roofline": roof,
    }

    if ctx.rank == 0 and ctx.world_size == 1:
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        if not args.no_extras:
            del ring
            torch.cuda.empty_cache()
            extras = {}
            extras["1e6_cache_hot"] = side_measurement(cfg, ctx.device, 1, spec["steps"], spec["warmup"])
            extras["1e5_cache_hot"] = side_measurement(synthetic.config2(), ctx.device, 1, 300, 30)
            extras["1e5_ring"] = side_measurement(synthetic.config2(), ctx.device, 64, 300, 30)
            extras["1e7_hbm"] = side_measurement(synthetic.config4(), ctx.device, 2, 50, 5)
            # the reference's own production size (examples/init-0.gsd stand-in): one single-block launch per evaluation
            extras["config1_N501"] = side_measurement(synthetic.config1(), ctx.device, 1, 2000, 100)
            extras["density_field_1e6_50k"] = density_field_measurement(cfg, ctx.device)
            extras["bussi_thermostat_step_1e6"] = thermostat_measurement(n, ctx.device)
            extras["1e6_energy_poll_every_step"] = energy_poll_measurement(cfg, ctx.device, spec["frames"], spec["steps"],
                                                                           spec["warmup"])
            extras["1e6_two_replicas_two_streams"] = two_replicas_measurement(cfg, ctx.device, spec["frames"], spec["steps"],
                                                                              spec["warmup"])
            line["extras"] = extras
    rank = ctx.rank
    replicas.shutdown(ctx)
    if rank == 0:
        # last thing on stdout (RCCL prints a version banner at init; nothing may follow the JSON line)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
