"""CPU-baseline worker for bench.py (TEST/BENCH INFRASTRUCTURE): runs the oracle's HMCDiag
(the NumPy restatement of bayes_kit/hmc.py) the way the reference runs -- one sampler
object per chain -- on a bounded slice of BASELINE.json config 3.  Imports numpy only, so
spawned workers never touch the GPU."""
import time

import numpy as np


def run_chains(chain0, nchains, draws, D, L, eps, seed):
    """Returns (leapfrog steps done, compute seconds)."""
    from oracle.models import DiagGaussian
    from oracle.samplers import HMCDiag

    lam = np.logspace(0, 4, D)
    samplers = []
    for c in range(chain0, chain0 + nchains):
        s = HMCDiag(DiagGaussian(lam), eps, L, metric_diag=np.ones(D), seed=np.random.Philox(key=[seed, c]))
        s._theta = s._theta / np.sqrt(lam)
        samplers.append(s)
    t0 = time.perf_counter()
    for s in samplers:
        for _ in range(draws):
            s.sample()
    return nchains * draws * L, time.perf_counter() - t0


if __name__ == "__main__":
    import json
    import sys

    a = sys.argv[1:]
    steps, secs = run_chains(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), int(a[6]))
    print(json.dumps({"steps": steps, "seconds": secs}))
