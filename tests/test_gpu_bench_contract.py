"""bench.py's contract, exercised on the GPU box: the single-GPU JSON line, and a two-rank rehearsal of the N > 1
launch (`torch.distributed.run`, both ranks on the one visible GPU, control collectives over gloo because RCCL refuses
two ranks on one device).  The real N = 2/4/8 runs over RCCL are the driver's to take."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5",
                          "--n-molecular", "200000", "--no-extras", "--cpu-seconds", "1"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in REQUIRED + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1.2
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # a plain single process joins no process group; its own rate is the whole job's
    assert d["config"]["dist_backend"] is None
    assert d["per_rank_evals_per_s"]["min"] == d["per_rank_evals_per_s"]["max"] >= d["value"] * (1 - 1e-9)


def test_two_rank_rehearsal():
    env = dict(os.environ, CAVMD_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
           "--warmup", "3", "--n-molecular", "200000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["replicas"] == 2 and d["config"]["collectives_on_data_path"] == 0
    # whole-job aggregate: two replicas' evaluations over the slowest rank's time
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    pr = d["per_rank_evals_per_s"]       # each rank's own K steps on its own clock: min <= max, both >= the whole-job per-rank rate
    assert 0 < pr["min"] <= pr["max"] and pr["min"] >= d["value"] / 2 * (1 - 1e-9)
    assert d["config"]["dist_backend"] == "gloo"


def test_plain_gpus2_form_self_launches():
    """The driver's command form: `python bench.py --gpus 2 ...` with NO outer torch.distributed.run.  bench.py starts
    its own two rank processes (before importing torch), relays rank 0's JSON line last and exits 0.  Both ranks share
    the one visible GPU here, so the control collectives run over gloo (RCCL refuses two ranks on one device)."""
    env = dict(os.environ, CAVMD_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
           "--n-molecular", "200000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1].startswith("{"), out.stdout[-500:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["replicas"] == 2 and d["config"]["collectives_on_data_path"] == 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
