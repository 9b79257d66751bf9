"""Config 4 at N = 2 against N = 1 on the same global chain ids.

    python tools/cfg4_two_rank_check.py [--chains-per-rank 32768] [--draws 40]

Runs `bench.py --gpus 2 --only cfg4` (2 x 32,768 chains; on a one-GPU box the ranks share the GPU and
rendezvous over gloo, on a two-GPU box they use RCCL) and `bench.py --gpus 1 --only cfg4 --chains 65536`,
then compares the R-hat of every dimension over the 65,536 global chains.  Chains are keyed by global id,
so both runs sample the same chains; the two R-hat vectors differ only by the order of the cross-chain sums
(per-rank partials combined in rank order vs one rank summing everything).  Prints one JSON record."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(extra, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only", "cfg4"] + extra, env=env,
                         capture_output=True, text=True, timeout=3000)
    if out.returncode != 0:
        sys.exit("bench.py failed:\n" + out.stderr[-3000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains-per-rank", type=int, default=32768)
    ap.add_argument("--draws", type=int, default=40)
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench

    share = {} if bench._visible_gpus() >= 2 else {"BK_BENCH_SHARE_GPU": "1"}
    two = run(["--gpus", "2", "--chains", str(a.chains_per_rank), "--steps", str(a.draws)], share)
    one = run(["--gpus", "1", "--chains", str(2 * a.chains_per_rank), "--steps", str(a.draws)], {})
    r2, r1 = two.pop("rhat"), one.pop("rhat")
    rel = max(abs(x / y - 1.0) for x, y in zip(r2, r1))
    rec = {"what": "R-hat of all 101 dims over the same 2 x %d global chains: 2 ranks (process group) vs 1 process"
                   % a.chains_per_rank,
           "max_rel_diff": rel, "ok_1e-12": rel <= 1e-12, "rhat_max_two_ranks": max(r2), "rhat_max_one_process": max(r1),
           "two_ranks": two, "one_process": one}
    print(json.dumps(rec))
    if rel > 1e-12:
        sys.exit(1)


if __name__ == "__main__":
    main()
