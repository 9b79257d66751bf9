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
