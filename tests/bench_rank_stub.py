"""Rank stub for the launcher test (tests/test_bench_contract.py): what a bench.py rank does
around its timed region -- rendezvous from the environment, a barrier, a max-over-ranks --
over gloo and without a GPU.  argv[1] = "ok" | "fail1" (rank 1 exits 3 before the rendezvous) | "cfg4" (what
bench.py's secondary.cfg4 does at world size N, on the CPU stand-in of the device ops: C chains per rank with global
chain ids rank*C.., R-hat over ALL ranks' chains through the process group, checked by rank 0 against one process
holding every chain)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
if sys.argv[1] == "fail1" and rank == 1:
    sys.exit(3)
if sys.argv[1] == "cfg4":
    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
    import bench

    pinned = bench.pin_rank(rank, world)  # (as a bench rank does, before the process group exists)
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
if sys.argv[1] == "cfg4":
    import numpy as np

    import bayes_kit_amd as bk
    from tests.fake_ops import FakeOps

    C, D, draws = 6, 5, 8
    args = (2, [0.5, 0.2], [2, 3], 0.4)
    ops = FakeOps()
    s = bk.DrGhmcDiag(bk.Funnel(D, ops=ops), *args, chains=C, chain_id0=rank * C, seed=20242, ops=ops)
    mom = bk.RunningMoments(D, C, ops=ops)
    for _ in range(draws):
        mom.update(s.sample()[0])
    rh = mom.rhat()                     # all_gather of the per-rank partial sums
    ids = [None] * world
    dist.all_gather_object(ids, (rank * C, C, sorted(os.sched_getaffinity(0))))
    if rank == 0:
        solo = dist.new_group(ranks=[0])
    else:
        dist.new_group(ranks=[0])
    if rank == 0:
        ref = bk.DrGhmcDiag(bk.Funnel(D, ops=ops), *args, chains=world * C, seed=20242, ops=ops)
        rmom = bk.RunningMoments(D, world * C, ops=ops)
        for _ in range(draws):
            rmom.update(ref.sample()[0])
        np.testing.assert_allclose(rh, rmom.rhat(group=solo), rtol=1e-12)
        cpus = [set(c) for _, _, c in ids]
        print(json.dumps({"n_gpus": world, "rhat_over_chains": sum(c for _, c, _ in ids),
                          "chain_id0": [f for f, _, _ in ids], "rhat_equals_one_process": True,
                          "affinity_disjoint": all(not (cpus[i] & cpus[j]) for i in range(world) for j in range(i)),
                          "affinity_sizes": [len(c) for c in cpus]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max_over_ranks": float(t.item()), "backend": dist.get_backend()}), flush=True)
else:
    print(f"rank {rank} done", flush=True)
dist.destroy_process_group()
