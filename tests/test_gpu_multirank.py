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
