"""RCCL on the GPU box, in the suite: what the N > 1 job does between ranks -- join with `device_id`, the 128-byte start-up
broadcast, barriers, MAX / MIN / SUM of a timing -- executed over backend "nccl" (= RCCL) with device-resident buffers by a
job of ONE rank on the box's one GPU (RCCL refuses two ranks on one device; the 2/4/8-GPU runs are the driver's).
Reference semantics of the N > 1 run: independent replicas, one process each (examples/05_advanced_run.py:1570-1612,
submit.sh:3): there is no data-path collective to test."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launcher_env():
    env = {k: v for k, v in os.environ.items() if k != "CAVMD_DIST_BACKEND"}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


CHILD = r'''
import os, sys
sys.path[:0] = [ROOT, os.path.join(ROOT, "cav-hoomd_amd")]
import torch, torch.distributed as dist
from cavitymd import replicas
ctx = replicas.init_from_env(prefer_gpu=True)
assert ctx.backend == "nccl" and ctx.is_distributed and dist.is_initialized() and dist.get_backend() == "nccl"
assert ctx.device.type == "cuda" and ctx.coll_device == ctx.device and dist.get_world_size() == 1
want = {"omegac": 0.00911267056242446, "couplstr": 1e-3, "phmass": 1.0, "n_molecular": 1000000, "base_seed": 0,
        "steps": 20, "warmup": 5, "frames": 7, "finite_q": True}
spec = replicas.broadcast_spec(ctx, want)                       # dist.broadcast of a float64[16] cuda tensor
assert all(spec[k] == v for k, v in want.items()), spec
replicas.barrier(ctx)
assert replicas.max_over_ranks(ctx, 1.25) == 1.25 and replicas.min_over_ranks(ctx, 2.5) == 2.5
assert replicas.sum_over_ranks(ctx, 3.75) == 3.75
# a collective on a bigger device buffer, checked on the device: the all-reduce of one rank is the identity
t = torch.arange(1 << 16, dtype=torch.float64, device=ctx.device)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
assert torch.equal(t, torch.arange(1 << 16, dtype=torch.float64, device=ctx.device))
replicas.barrier(ctx)
replicas.shutdown(ctx)
assert not dist.is_initialized()
print("RCCL-WORLD1-OK", torch.cuda.nccl.version())
'''


def test_rccl_world1_control_collectives_on_device_buffers():
    out = subprocess.run([sys.executable, "-c", CHILD.replace("ROOT", repr(ROOT))], capture_output=True, text=True, timeout=600,
                         env=_launcher_env(), cwd=ROOT)
    assert out.returncode == 0 and "RCCL-WORLD1-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_line_of_a_launched_one_rank_job_runs_over_rccl():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launcher form of the driver's command with N = 1.
    bench.py joins the job, so its broadcast / barrier / MAX / MIN go through RCCL; the JSON line says which backend ran."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--n-molecular", "200000", "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("CAVMD_DIST_BACKEND", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["dist_backend"] == "nccl" and d["config"]["dist_collective_device"].startswith("cuda")
    pr = d["per_rank_evals_per_s"]
    assert pr["min"] == pr["max"] and pr["min"] >= d["value"] * (1 - 1e-9)   # own clock stops before the closing barrier
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
