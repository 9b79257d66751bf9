"""One rank of the multi-rank GPU test (tests/test_gpu_multirank.py): sharded DRGHMC chains on this
rank's GPU, R-hat over ALL ranks' chains through the process group (bayes_kit/rhat.py:163-171), checked on
rank 0 against one process holding every chain.  BK_TEST_BACKEND = nccl (one GPU per rank: RCCL over xGMI)
or gloo (the ranks share GPU 0: a code-path exercise for one-GPU boxes)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]

import numpy as np
import torch
import torch.distributed as dist

import bayes_kit_amd as bk


def main():
    import faulthandler

    # a rank stuck in a collective says where (every thread's stack) and exits, instead of hanging the test
    faulthandler.dump_traceback_later(float(os.environ.get("BK_TEST_WATCHDOG", "150")), exit=True)
    backend = os.environ.get("BK_TEST_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ["LOCAL_RANK"]) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    assert dist.get_backend() == backend
    D, C_total, N, seed = 21, 1001, 25, 77  # an odd total: uneven shards
    args = (3, [0.3, 0.1, 0.03], [3, 9, 27], 0.3)
    first, n = bk.dist.shard(C_total)
    s = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=n, chain_id0=first, seed=seed)
    mom = bk.RunningMoments(D, n)
    rec = bk.DrawRecorder([0, D - 1], N, n)
    for _ in range(N):
        th, lp = s.sample()
        mom.update(th)
        rec.record(th, lp)
    calls0 = dict(bk.dist.collective_calls)
    rh = mom.rhat()  # two all_gathers of 3 D + 1 doubles
    assert bk.dist.collective_calls["all_gather"] - calls0["all_gather"] == 2
    ess_sum = bk.dist.sum_over_ranks(float(rec.ess().clamp(min=0.0, max=float(N)).sum().item()), dev)
    # rank-normalised R-hat of one coordinate: global ranks by the cross-rank sample sort (all_to_all)
    rn = bk.rank_normalized_rhat(rec.series[0, :N])
    # every rank got the same numbers
    box = [None] * world
    dist.all_gather_object(box, (rh.tolist(), ess_sum, float(rn)))
    assert all(b == box[0] for b in box), "ranks disagree"
    # one process holding every chain (a one-rank subgroup keeps its summaries local)
    solo = [dist.new_group(ranks=[r]) for r in range(world)][rank]
    if rank == 0:
        ref = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=C_total, seed=seed)
        rmom = bk.RunningMoments(D, C_total)
        rrec = bk.DrawRecorder([0, D - 1], N, C_total)
        for _ in range(N):
            th, lp = ref.sample()
            rmom.update(th)
            rrec.record(th, lp)
        np.testing.assert_allclose(rh, rmom.rhat(group=solo), rtol=1e-12)
        want_ess = float(rrec.ess().clamp(min=0.0, max=float(N)).sum().item())
        np.testing.assert_allclose(ess_sum, want_ess, rtol=1e-12)
        np.testing.assert_allclose(rn, bk.rank_normalized_rhat(rrec.series[0, :N], group=solo), rtol=1e-12)
        assert torch.equal(rrec.series[0, :N, first:first + n], rec.series[0, :N])  # my shard, bit for bit
    # logistic regression (two MFMA GEMMs per gradient, X^T r split over the observations) under HMC with a dense
    # metric: the split is a function of (D, N) only, so a shard's chains are the one-process run's, bit for bit
    gen = torch.Generator().manual_seed(5)
    Nl, Dl, Cl = 20_000, 32, 301
    X = (torch.randn((Nl, Dl), dtype=torch.float64, generator=gen) / Dl ** 0.5).to(dev)
    y = (torch.rand(Nl, dtype=torch.float64, generator=gen) < 0.5).to(torch.float64).to(dev)
    Md = torch.eye(Dl, dtype=torch.float64) * 0.02 + 0.001
    f2, n2 = bk.dist.shard(Cl)
    sl = bk.HMCDiag(bk.LogisticRegression(X, y), 0.1, 3, chains=n2, chain_id0=f2, seed=seed, metric_dense=Md)
    mine = [sl.sample() for _ in range(3)]
    if rank == 0:
        rl = bk.HMCDiag(bk.LogisticRegression(X, y), 0.1, 3, chains=Cl, seed=seed, metric_dense=Md)
        for n_, (t_, l_) in enumerate(mine):
            tr, lr = rl.sample()
            assert torch.equal(tr[f2:f2 + n2], t_) and torch.equal(lr[f2:f2 + n2], l_), ("logistic shard", n_)
    dist.barrier()
    dist.destroy_process_group()
    import json

    print(json.dumps({"rank": rank, "ok": True, "backend": backend, "gpu": local}), flush=True)


if __name__ == "__main__":
    main()
