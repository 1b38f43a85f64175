"""Exercises the RCCL calls bench.py makes for N > 1 (init with device_id, broadcast, all_reduce MAX/SUM, barrier) with a
world size of 1 on the one GPU a gpurun box has.  The real multi-GPU runs are the driver's."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "cav-hoomd_amd")]
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch, torch.distributed as dist
from cavitymd import replicas
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
ctx = replicas.ReplicaContext(0, 2, 0, "nccl", dev, dev)   # world_size 2 in the context so the collective branches run
object.__setattr__(ctx, "world_size", 2)
spec = replicas.broadcast_spec(ctx, {"omegac": 0.0091, "couplstr": 1e-3, "phmass": 1.0, "n_molecular": 1000, "base_seed": 0,
                                     "steps": 3, "warmup": 1, "frames": 2, "finite_q": True})
replicas.barrier(ctx)
print("spec", spec["n_molecular"], "max", replicas.max_over_ranks(ctx, 1.5), "sum", replicas.sum_over_ranks(ctx, 2.0))
dist.destroy_process_group()
print("rccl sanity ok")
