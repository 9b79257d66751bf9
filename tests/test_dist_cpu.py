"""N>1 path on CPU: two gloo ranks (fake ops), chain sharding + cross-rank R-hat."""
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharding_and_rhat():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


def test_shard_arithmetic():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
    import bayes_kit_amd as bk

    for total, world in [(262144, 8), (10, 3), (7, 8), (0, 2)]:
        blocks = [bk.dist.shard(total, r, world) for r in range(world)]
        assert sum(n for _, n in blocks) == total
        pos = 0
        for first, n in blocks:
            assert first == pos
            pos += n


def test_single_rank_group_with_forced_collectives_over_gloo():
    """The code path tests/test_gpu_multirank.py runs on RCCL, here on gloo: a one-rank group with the summaries'
    collectives forced through it gives the no-group answers (all_gather lists, all_to_all_single with count lists,
    the rank-normalised R-hat's sample sort with zero splitters)."""
    import json
    import socket
    import subprocess
    import sys

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"BK_TEST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "OMP_NUM_THREADS": "1"})
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "single_rank_group_worker.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert r["ok"] and r["backend"] == "gloo" and r["collectives"]["all_to_all"] >= 5
