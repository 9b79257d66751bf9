"""bench.py contract pieces that do not need a GPU: CLI, the CPU-baseline leg, traffic file."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_cli_help_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config"):
        assert flag in out.stdout


def test_cpu_baseline_worker_is_the_oracle_and_reports_steps():
    # one tiny worker run: the thing bench.py times as `cpu_baseline` (kind "port")
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "0", "2", "3", "32", "4", "0.05", "7"], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["steps"] == 2 * 3 * 4 and r["seconds"] > 0


def test_committed_traffic_matches_the_algorithmic_bytes():
    sys.path.insert(0, ROOT)
    import bench

    t = bench._pmc_traffic(65536, 1024)
    assert t is not None
    algorithmic = 40.0 * 1024 * 65536
    assert abs(t / algorithmic - 1.0) < 0.02  # PMC traffic == algorithmic bytes (no re-reads)
    assert bench._pmc_traffic(4096, 128) is None
