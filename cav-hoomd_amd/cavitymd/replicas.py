"""Independent replicas, one process per GPU.

The reference runs ``--replicas 1-8`` as a sequential loop in one process, or as one SLURM array task
per replica (examples/05_advanced_run.py:1336-1351, 1570-1612; submit.sh:3).  Replicas never exchange
data, so here they are sharded over the ranks of a ``torch.distributed`` job (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) with NO collective on the data path.  The only communication
is one start-up broadcast of a small parameter block (seeds, cavity parameters, sizes) from rank 0, so
that every rank provably runs the same experiment definition.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist

# layout of the broadcast block (float64 x 16); integers are exact in a double up to 2^53
_FIELDS = ("omegac", "couplstr", "phmass", "n_molecular", "base_seed", "steps", "warmup", "frames", "finite_q",
           "version")
BLOCK_DOUBLES = 16
BLOCK_VERSION = 1.0


def parse_replicas(spec: str | None):
    """'1-8' / '1,3,5' / '1-3,7' -> sorted unique replica ids; empty -> [1] (same grammar as the
    reference's --replicas flag, examples/05_advanced_run.py:1336-1351)."""
    if not spec:
        return [1]
    ids = set()
    for token in spec.split(","):
        token = token.strip()
        if not token:
            continue
        if "-" in token:
            lo, hi = token.split("-", 1)
            ids.update(range(int(lo), int(hi) + 1))
        else:
            ids.add(int(token))
    return sorted(ids)


def assign_replicas(replica_ids, world_size: int):
    """Round-robin: the k-th replica of the sorted list goes to rank k mod world_size."""
    plan = [[] for _ in range(world_size)]
    for k, rid in enumerate(sorted(replica_ids)):
        plan[k % world_size].append(rid)
    return plan


def replica_seed(replica_id: int, base_seed: int = 0) -> int:
    """Seed of a replica.  BASELINE config 5 uses seeds 1-8 for replicas 1-8 (base_seed 0).  (The reference
    draws its HOOMD seed from np.random.randint and is not reproducible, examples/05_advanced_run.py:401.)"""
    return int(base_seed) + int(replica_id)


@dataclass
class ReplicaContext:
    rank: int
    world_size: int
    local_rank: int
    backend: str | None
    device: torch.device            # where the replica computes
    coll_device: torch.device = torch.device("cpu")  # where the (tiny) collective buffers live

    @property
    def is_distributed(self) -> bool:
        """True when this process has joined a torch.distributed job -- also a job of ONE rank (torch.distributed.run
        --nproc-per-node 1): the collectives then really run (over RCCL on a GPU box), which is how the one-GPU test box
        exercises the N > 1 code path's backend calls."""
        return self.backend is not None


def init_from_env(prefer_gpu: bool = True) -> ReplicaContext:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and join the job if there is one."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = prefer_gpu and torch.cuda.is_available()
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1)) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    backend = None
    coll_device = torch.device("cpu")
    # a launcher's environment (WORLD_SIZE and MASTER_ADDR both set) means "join the job", whatever its size; a plain
    # `python bench.py` has neither and stays a single process without a process group
    launched = "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or launched:
        # nccl == RCCL over xGMI on the GPU box.  CAVMD_DIST_BACKEND=gloo keeps the compute on the GPU but runs the
        # handful of control collectives over gloo: used to rehearse an N-rank job on a box with fewer GPUs than ranks
        # (RCCL refuses two ranks on one device).
        backend = os.environ.get("CAVMD_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            coll_device = device
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kwargs = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return ReplicaContext(rank, world, local, backend, device, coll_device)


def pack_block(spec: dict) -> torch.Tensor:
    t = torch.zeros(BLOCK_DOUBLES, dtype=torch.float64)
    for k, name in enumerate(_FIELDS):
        t[k] = float(BLOCK_VERSION if name == "version" else spec.get(name, 0.0))
    return t


def unpack_block(t: torch.Tensor) -> dict:
    vals = t.detach().cpu().tolist()
    out = {name: vals[k] for k, name in enumerate(_FIELDS)}
    for name in ("n_molecular", "base_seed", "steps", "warmup", "frames"):
        out[name] = int(round(out[name]))
    out["finite_q"] = bool(round(out["finite_q"]))
    if out["version"] != BLOCK_VERSION:
        raise RuntimeError(f"parameter block version mismatch: got {out['version']}, expected {BLOCK_VERSION}")
    return out


def broadcast_spec(ctx: ReplicaContext, spec: dict | None) -> dict:
    """Rank 0's experiment definition, delivered to every rank (128 bytes, one collective, start-up only)."""
    if not ctx.is_distributed:
        if spec is None:
            raise ValueError("rank 0 must supply the spec")
        return unpack_block(pack_block(spec))
    block = pack_block(spec if (ctx.rank == 0 and spec is not None) else {})
    block = block.to(ctx.coll_device)
    dist.broadcast(block, src=0)
    return unpack_block(block)


def barrier(ctx: ReplicaContext) -> None:
    if ctx.is_distributed:
        dist.barrier()


def max_over_ranks(ctx: ReplicaContext, value: float) -> float:
    if not ctx.is_distributed:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=ctx.coll_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(ctx: ReplicaContext, value: float) -> float:
    if not ctx.is_distributed:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=ctx.coll_device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def sum_over_ranks(ctx: ReplicaContext, value: float) -> float:
    if not ctx.is_distributed:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=ctx.coll_device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown(ctx: ReplicaContext) -> None:
    if ctx.is_distributed and dist.is_initialized():
        dist.destroy_process_group()
