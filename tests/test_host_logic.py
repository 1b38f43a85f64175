"""Host-side logic that needs no GPU: the Python surface mirrored from the reference, the particle-data
stand-in, unit constants, synthetic configurations, replica planning."""
import json
import os

import numpy as np
import pytest
import torch

import cavitymd
from cavitymd import replicas, synthetic
from cavitymd.state import type_tag_as_double


def test_package_exports_reference_names():
    # src/cavitymd/__init__.py:6-13 of the reference exports these for the force path
    for name in ("CavityForce", "PhysicalConstants", "unwrap_positions"):
        assert hasattr(cavitymd, name)


def test_constants_and_conversions_match_reference_utils(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "utils_golden.json")))
    PC = cavitymd.PhysicalConstants
    for k, v in g["constants"].items():
        assert getattr(PC, k) == v, k
    for t, v in g["conversions"]["ps_to_atomic_units"]:
        assert PC.ps_to_atomic_units(t) == v
    for t, v in g["conversions"]["atomic_units_to_ps"]:
        assert PC.atomic_units_to_ps(t) == v
    for t, v in g["conversions"]["gamma_from_tau_ps"]:
        assert PC.gamma_from_tau_ps(t) == v
    with pytest.raises(ValueError):
        PC.gamma_from_tau_ps(0.0)
    assert PC.omegac_from_wavenumber(2000.0) == g["omegac_2000cm"]
    for case in g["unwrap_cases"]:
        got = cavitymd.unwrap_positions(np.array(case["positions"]), np.array(case["images"]), np.array(case["box"]))
        assert np.array_equal(got, np.array(case["unwrapped"]))


def test_cavity_force_surface():
    f = cavitymd.CavityForce(kvector=[0, 0, 1], couplstr=1e-3, omegac=0.00911267, phmass=1.0)
    assert np.array_equal(f.kvector, np.array([0.0, 0.0, 1.0]))
    assert (f.couplstr, f.omegac, f.phmass) == (1e-3, 0.00911267, 1.0)
    # before attach every loggable reads 0.0, as the reference's `if self._force_impl else 0.0` (forces.py:183-198)
    assert f.harmonic_energy == 0.0 and f.coupling_energy == 0.0 and f.dipole_self_energy == 0.0
    assert f.total_cavity_energy == 0.0 and f.energy == 0.0
    assert f.forces is None and f.set_forces(0) is None
    with pytest.raises(RuntimeError):
        f.compute(0)
    with pytest.raises(ValueError):
        cavitymd.CavityForce(kvector=[0, 1], couplstr=1e-3, omegac=0.1)


def test_particle_data_accessors():
    cfg = synthetic.config1()
    pd = cavitymd.ParticleData.from_arrays(cfg["position"], cfg["typeid"], cfg["charge"], cfg["image"], cfg["types"],
                                           cfg["box"], device="cpu")
    assert pd.getN() == 501
    assert pd.getTypeByName("L") == 2 and pd.getTypeByName("O") == 0
    with pytest.raises(RuntimeError, match="Type X not found"):
        pd.getTypeByName("X")
    assert pd.getGlobalBox().getL() == (40.0, 40.0, 40.0)
    pos = pd.getPositions().numpy()
    # the type id sits in the low 32 bits of pos.w
    tags = (pos[:, 3].view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(tags, cfg["typeid"])
    assert type_tag_as_double(np.array([-1])).view(np.uint64)[0] == 0xFFFFFFFF
    with pytest.raises(ValueError):
        cavitymd.ParticleData(torch.zeros(3, 3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64),
                              torch.zeros(3, 3, dtype=torch.int32), ["A"], (1, 1, 1))


def test_synthetic_configs_follow_the_schema():
    c1 = synthetic.config1()
    assert c1["position"].shape == (501, 3) and c1["types"] == ["O", "N", "L"]
    L = c1["box"][0]
    assert np.all(c1["position"] >= -L / 2) and np.all(c1["position"] < L / 2)
    assert abs(c1["charge"][:-1].sum()) < 1e-12          # neutral molecules
    assert set(np.unique(c1["image"][:-1])) <= {-2, -1, 0, 1, 2}
    # bond lengths survive the wrap: O-O 2.2817, N-N 2.0744 bohr
    r = c1["position"] + c1["image"] * np.asarray(c1["box"])[None, :]
    bonds = np.linalg.norm(r[0:500:2] - r[1:500:2], axis=1)
    assert np.allclose(bonds[0::2], synthetic.BOND_OO) and np.allclose(bonds[1::2], synthetic.BOND_NN)
    c2 = synthetic.config2(n_molecular=2000)
    assert c2["position"].shape == (2001, 3) and abs(c2["charge"].sum()) < 1e-9
    assert np.array_equal(c2["typeid"][:-1], np.arange(2000) % 2) and c2["typeid"][-1] == 2
    # finite-q: photon sits near -d g / omegac^2 (z about 0), examples/05_advanced_run.py:464-471
    c3 = synthetic.config3(n_molecular=2000)
    r3 = c3["position"] + c3["image"] * np.asarray(c3["box"])[None, :]
    d = (c3["charge"][:-1, None] * r3[:-1]).sum(axis=0)
    p = c3["params"]
    sigma = np.sqrt(cavitymd.PhysicalConstants.KB_HARTREE_PER_K * 100.0 / p["omegac"]**2)
    target = -d * p["couplstr"] / p["omegac"]**2
    assert abs(r3[-1, 0] - target[0]) < 6 * sigma and abs(r3[-1, 1] - target[1]) < 6 * sigma and abs(r3[-1, 2]) < 6 * sigma
    # replicas differ only by seed
    a, b = synthetic.config5_replica(0, 200), synthetic.config5_replica(1, 200)
    assert a["seed"] == 1 and b["seed"] == 2 and not np.array_equal(a["position"], b["position"])
    # pseudo-trajectory keeps particles wrapped and moves them a little
    s = synthetic.perturb(c1, 7)
    r1 = s["position"] + s["image"] * np.asarray(s["box"])[None, :]
    assert np.all(s["position"] >= -L / 2) and np.all(s["position"] < L / 2)
    assert 0 < np.abs(r1 - r).max() < 1e-2


def test_replica_planning():
    assert replicas.parse_replicas("1-8") == [1, 2, 3, 4, 5, 6, 7, 8]
    assert replicas.parse_replicas("3,1, 1-2,7") == [1, 2, 3, 7]
    assert replicas.parse_replicas("") == [1] and replicas.parse_replicas(None) == [1]
    plan = replicas.assign_replicas(range(1, 9), 4)
    assert plan == [[1, 5], [2, 6], [3, 7], [4, 8]]
    assert sorted(sum(replicas.assign_replicas([1, 2, 3], 8), [])) == [1, 2, 3]
    assert [replicas.replica_seed(r) for r in range(1, 9)] == list(range(1, 9))
    spec = {"omegac": 0.0091, "couplstr": 1e-3, "phmass": 1.0, "n_molecular": 1_000_000, "base_seed": 0, "steps": 50,
            "warmup": 5, "frames": 7, "finite_q": True}
    back = replicas.unpack_block(replicas.pack_block(spec))
    for k, v in spec.items():
        assert back[k] == v
    bad = replicas.pack_block(spec)
    bad[9] = 99.0
    with pytest.raises(RuntimeError, match="version"):
        replicas.unpack_block(bad)


def test_bench_self_launch_plumbing_without_a_gpu():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two rank processes itself and returns their worst
    exit code.  Without a GPU every rank refuses to run (the product has no CPU path), which is exactly what makes the
    plumbing observable here: two refusals on stderr, a non-zero exit code, no JSON line."""
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CAVMD_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--n-molecular", "1000"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("{")
    else:
        assert out.returncode != 0
        assert out.stderr.count("bench.py needs a GPU") == 2, out.stderr[-2000:]
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_launch_ranks_hands_out_eight_ranks_and_relays_rank0_json_last(tmp_path, capfd):
    """The driver's N = 8 form, `python bench.py --gpus 8`, without a GPU: the parent starts eight processes with
    RANK = LOCAL_RANK = 0..7, WORLD_SIZE = LOCAL_WORLD_SIZE = 8, ONE rendezvous address 127.0.0.1:<port> for all, the dmabuf
    IPC switch set; whatever the ranks print, rank 0's JSON line is the last thing on stdout; the exit code is the worst."""
    import json
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import json, os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "json.dump({k: os.environ.get(k) for k in keys} | {'argv': sys.argv[1:]}, open(os.path.join(sys.argv[1], f'env{r}.json'), 'w'))\n"
        "if r == 0:\n"
        "    print('NCCL version banner'); print(json.dumps({'metric': 'm', 'value': 1.0})); print('trailing chatter')\n"
        "else:\n"
        "    time.sleep(0.05 * r); print(f'rank {r} says hello')\n"
        "sys.exit(int(sys.argv[2]) if r == 5 else 0)\n")
    sys.path.insert(0, root)
    import bench
    rc = bench.launch_ranks(8, [str(tmp_path), "0"], script=str(probe))
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip().splitlines() == [json.dumps({"metric": "m", "value": 1.0})]       # and nothing else on stdout
    assert "NCCL version banner" in err and "trailing chatter" in err and "rank 7 says hello" in err
    envs = [json.load(open(tmp_path / f"env{r}.json")) for r in range(8)]
    assert [e["RANK"] for e in envs] == [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(8)]
    assert {e["WORLD_SIZE"] for e in envs} == {e["LOCAL_WORLD_SIZE"] for e in envs} == {"8"}
    assert {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"} and len({e["MASTER_PORT"] for e in envs}) == 1
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}
    assert all(e["argv"] == [str(tmp_path), "0"] for e in envs)
    # one failing rank: its exit code comes back (the others have finished by themselves here)
    assert bench.launch_ranks(8, [str(tmp_path), "3"], script=str(probe)) == 3
    capfd.readouterr()


def test_a_launched_job_of_one_rank_joins_a_process_group(tmp_path):
    """WORLD_SIZE=1 from a launcher still means `join`: the collectives run (gloo here, RCCL on the GPU box) instead of being
    skipped, so a one-GPU box exercises the same calls as the 8-GPU job.  A plain process (no launcher variables) joins nothing."""
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    code = ("import os, sys\n"
            f"sys.path[:0] = [{root!r}, os.path.join({root!r}, 'cav-hoomd_amd')]\n"
            "import torch.distributed as dist\n"
            "from cavitymd import replicas\n"
            "ctx = replicas.init_from_env(prefer_gpu=False)\n"
            "print('JOINED', ctx.is_distributed, ctx.backend, dist.is_initialized())\n"
            "spec = replicas.broadcast_spec(ctx, {'omegac': 0.5, 'n_molecular': 12, 'steps': 3})\n"
            "replicas.barrier(ctx)\n"
            "print('VALUES', spec['n_molecular'], replicas.max_over_ranks(ctx, 1.5), replicas.min_over_ranks(ctx, 2.5), replicas.sum_over_ranks(ctx, 3.5))\n"
            "replicas.shutdown(ctx)\n")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=base)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "JOINED False None False" in out.stdout and "VALUES 12 1.5 2.5 3.5" in out.stdout
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(base, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "JOINED True gloo True" in out.stdout and "VALUES 12 1.5 2.5 3.5" in out.stdout
