"""The N > 1 path on CPU: world_size-2 `gloo` job exercising exactly what bench.py does across ranks --
join from the environment, broadcast the parameter block from rank 0, shard replicas (no data-path collective),
barrier, max/sum over ranks.  The replicas here evaluate the ORACLE instead of the HIP kernels (no GPU in this
container); the point is the distributed plumbing, not the force."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "cav-hoomd_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import oracle
    from cavitymd import replicas, synthetic
    ctx = replicas.init_from_env(prefer_gpu=False)
    assert ctx.backend == "gloo" and ctx.world_size == world and ctx.rank == rank
    spec = None
    if rank == 0:
        p = synthetic.default_params()
        spec = {"omegac": p["omegac"], "couplstr": p["couplstr"], "phmass": p["phmass"], "n_molecular": 2000,
                "base_seed": 0, "steps": 3, "warmup": 1, "frames": 2, "finite_q": True}
    spec = replicas.broadcast_spec(ctx, spec)  # non-zero ranks pass None and must receive rank 0's block
    mine = replicas.assign_replicas(replicas.parse_replicas("1-4"), world)[rank]
    ref = oracle.RefOracle()
    energies = {}
    for rid in mine:
        cfg = synthetic.diatomic_box(spec["n_molecular"], seed=replicas.replica_seed(rid, spec["base_seed"]),
                                     finite_q=spec["finite_q"],
                                     params={k: spec[k] for k in ("omegac", "couplstr", "phmass")})
        prm = ref.make_params(spec["omegac"], spec["couplstr"], spec["phmass"])
        out = ref.compute(oracle.pack_pos(cfg["position"], cfg["typeid"]), cfg["charge"], cfg["image"], cfg["box"],
                          cfg["L_typeid"], prm)
        energies[rid] = out["energies"]
    replicas.barrier(ctx)
    slowest = replicas.max_over_ranks(ctx, float(rank + 1))
    total = replicas.sum_over_ranks(ctx, float(len(mine)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=np.array(mine), e=np.array([energies[r] for r in mine]),
             slowest=slowest, total=total, n_molecular=spec["n_molecular"], omegac=spec["omegac"])
    replicas.shutdown(ctx)


def test_two_rank_gloo_replica_job(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert r0["ids"].tolist() == [1, 3] and r1["ids"].tolist() == [2, 4]
    assert r0["slowest"] == r1["slowest"] == 2.0 and r0["total"] == r1["total"] == 4.0
    # rank 1 received rank 0's parameter block
    assert r1["n_molecular"] == 2000 and r1["omegac"] == r0["omegac"] == 2000.0 / 219474.63
    # replicas are independent: different seeds, different energies; same seed would reproduce
    assert not np.allclose(r0["e"][0], r1["e"][0])
