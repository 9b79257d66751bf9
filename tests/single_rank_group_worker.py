"""A process group of ONE rank with the collectives forced through it (bayes_kit_amd.dist.force_collectives): every collective
the N > 1 summaries issue -- all_gather with a tensor list, all_to_all_single with count lists, all_reduce, the rank-normalised
R-hat's sample sort -- runs on the backend named in BK_TEST_BACKEND (nccl = RCCL on the GPU box, gloo on the build box) and must
give what the no-group path gives.  Started as a child process by the tests (the parent may have touched the GPU)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    import faulthandler

    faulthandler.dump_traceback_later(float(os.environ.get("BK_TEST_WATCHDOG", "150")), exit=True)
    backend = os.environ.get("BK_TEST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    if backend == "nccl":
        import bayes_kit_amd as bk

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        ops = None
    else:
        from tests.fake_ops import FakeOps  # the CPU stand-in of the device ops (test infrastructure)

        import bayes_kit_amd as bk

        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=0, world_size=1)
        ops = FakeOps()
    assert dist.get_backend() == backend and dist.get_world_size() == 1
    kw = {} if ops is None else {"ops": ops}
    g = torch.Generator().manual_seed(11)
    x = torch.randn((200, 37), dtype=torch.float64, generator=g)
    x[:, 5] += 0.4
    x[50:60, 7] = x[40:50, 7]  # some ties
    x = x.to(dev)
    # the no-group answers first (collectives not forced)
    bk.dist.force_collectives = False
    want = {"rhat": float(bk.rhat(x, **kw)), "split": float(bk.split_rhat(x, **kw)), "rank": float(bk.rank_normalized_rhat(x, **kw))}
    assert bk.dist.gather_sum(x) is x
    calls0 = dict(bk.dist.collective_calls)
    bk.dist.force_collectives = True
    got = {"rhat": float(bk.rhat(x, **kw)), "split": float(bk.split_rhat(x, **kw)), "rank": float(bk.rank_normalized_rhat(x, **kw))}
    calls = {k: bk.dist.collective_calls[k] - calls0[k] for k in calls0}
    assert calls["all_gather"] >= 6 and calls["all_to_all"] >= 5, calls  # (the sample sort: counts, keys, payload, and back)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-12, err_msg=k)
    # the primitives on their own
    parts = bk.dist.all_gather(x)
    assert len(parts) == 1 and torch.equal(parts[0], x)
    s = bk.dist.gather_sum(x)
    assert s is not x and torch.equal(s, x)
    recv = torch.empty(200 * 37, dtype=torch.float64, device=dev)
    bk.dist.all_to_all_single(recv, x.reshape(-1), [200 * 37], [200 * 37])
    assert torch.equal(recv, x.reshape(-1))
    assert bk.dist.sum_over_ranks(3.25, dev) == 3.25
    if backend == "nccl":
        # Welford moments + recorder of a sharded sampler through the forced group == the local summaries
        s_ = bk.DrGhmcDiag(bk.Funnel(21), 3, [0.3, 0.1, 0.03], [3, 9, 27], 0.3, chains=500, seed=5)
        mom, rec = bk.RunningMoments(21, 500), bk.DrawRecorder([0, 20], 20, 500)
        for _ in range(20):
            th, lp = s_.sample()
            mom.update(th)
            rec.record(th, lp)
        forced = mom.rhat()
        bk.dist.force_collectives = False
        np.testing.assert_allclose(np.asarray(forced), np.asarray(mom.rhat()), rtol=1e-12)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "backend": backend, "collectives": calls}), flush=True)


if __name__ == "__main__":
    main()
