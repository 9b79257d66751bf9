"""N > 1 on the GPU.  Two rank processes are SPAWNED (never re-exec'ed from a process that
touched the GPU): over RCCL (`nccl`) when the box has two GPUs -- skipped otherwise, so it runs by itself
the day such a box appears -- and over gloo with both ranks on GPU 0, which exercises the same sharding,
R-hat / ESS / rank-normalised R-hat code on every box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def _run_ranks(backend, world=2):
    import io

    import bench

    buf = io.StringIO()
    rc = bench.launch_ranks(world, [sys.executable, os.path.join(ROOT, "tests", "rank_worker_gpu.py")],
                            {"BK_TEST_BACKEND": backend, "OMP_NUM_THREADS": "1"}, timeout=240, out=buf)
    assert rc == 0, buf.getvalue()
    r0 = json.loads(buf.getvalue().strip().splitlines()[-1])  # (the launcher relays rank 0's JSON lines)
    assert r0["ok"] and r0["rank"] == 0 and r0["backend"] == backend


def test_two_ranks_share_one_gpu_over_gloo():
    _run_ranks("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_over_rccl():
    _run_ranks("nccl")


def test_bench_two_ranks_config4_rhat_equals_one_process():
    """bench.py's N > 1 leg does what north_star's config 4 says: chains sharded by global id, R-hat over
    all ranks' chains through the process group.  2 ranks x 2,048 chains == 1 process x 4,096 chains."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", "cfg4", "--steps", "12"]

    def run(extra, env_extra):
        out = subprocess.run(base + extra, env=dict(env, **env_extra), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    two = run(["--gpus", "2", "--chains", "2048"], {"BK_BENCH_SHARE_GPU": "1"} if torch.cuda.device_count() < 2 else {})
    one = run(["--gpus", "1", "--chains", "4096"], {})
    assert two["n_gpus"] == 2 and two["rhat_over_chains"] == 4096 == one["rhat_over_chains"]
    assert two["collectives_per_summary"] == {"all_gather": 2, "all_reduce": 2, "all_to_all": 0}
    assert two["collective_ranks"] == 2 and one["collective_ranks"] == 0
    np.testing.assert_allclose(two["rhat"], one["rhat"], rtol=1e-12)
    assert abs(two["mean_grad_evals_per_draw"] - one["mean_grad_evals_per_draw"]) < 1e-9


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_rank_rccl_group_runs_every_collective_of_the_summaries():
    """First contact with RCCL without a second GPU: a ONE-rank `nccl` group on this GPU with the
    summaries' collectives forced through it -- dist.all_gather with a tensor list, all_to_all_single with count lists,
    gather_sum, all_reduce and the rank-normalised R-hat's sample sort -- gives the no-group answers."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"BK_TEST_BACKEND": "nccl", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "single_rank_group_worker.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    # (RCCL prints a version banner on the C side of stdout, flushed at exit: the JSON line is not the last one)
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["ok"] and r["backend"] == "nccl" and r["collectives"]["all_to_all"] >= 5


def test_bench_single_rank_on_rccl():
    """bench.py as ONE rank of a torch.distributed.run-style launch (WORLD_SIZE=1 in the environment) with the `nccl`
    group forced (BK_BENCH_FORCE_GROUP=1): config 4's summary -- R-hat all_gathers, ESS / lane all_reduces -- goes through
    RCCL and equals the plain single-process run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", "cfg4", "--steps", "12", "--chains", "2048", "--gpus", "1"]

    def run(env_extra):
        out = subprocess.run(base, env=dict(env, **env_extra), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    forced = run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                  "BK_BENCH_FORCE_GROUP": "1", "BK_DIST_FORCE_COLLECTIVES": "1"})
    plain = run({})
    assert forced["collective_backend"] == "nccl" and forced["collective_ranks"] == 1
    assert forced["collectives_per_summary"] == {"all_gather": 2, "all_reduce": 2, "all_to_all": 0}
    assert plain["collective_backend"] is None and plain["collectives_per_summary"]["all_gather"] == 0
    np.testing.assert_allclose(forced["rhat"], plain["rhat"], rtol=1e-12)
    assert abs(forced["mean_grad_evals_per_draw"] - plain["mean_grad_evals_per_draw"]) < 1e-9
