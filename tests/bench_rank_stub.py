"""Rank stub for the launcher test (tests/test_bench_contract.py): what a bench.py rank does
around its timed region -- rendezvous from the environment, a barrier, a max-over-ranks --
over gloo and without a GPU.  argv[1] = "ok" | "fail1" (rank 1 exits 3 before the rendezvous)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
if sys.argv[1] == "fail1" and rank == 1:
    sys.exit(3)
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max_over_ranks": float(t.item()), "backend": dist.get_backend()}), flush=True)
else:
    print(f"rank {rank} done", flush=True)
dist.destroy_process_group()
