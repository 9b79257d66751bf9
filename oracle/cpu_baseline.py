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


def run_batched(nchains, draws, D, L, eps, seed):
    """Context figure: the same HMC written the way a NumPy user would vectorise it
    by hand -- all chains of a process in [D, C] arrays, in-place ufuncs, no temporaries -- NOT
    the reference's execution model.  Same arithmetic per element as HMCDiag.sample().
    Returns (leapfrog steps done, compute seconds)."""
    rng = np.random.default_rng(seed)
    lam = np.logspace(0, 4, D)[:, None]
    m = np.ones((D, 1))
    theta = rng.normal(size=(D, nchains)) / np.sqrt(lam)
    g, t, r, prop = (np.empty((D, nchains)) for _ in range(4))
    half = 0.5 * eps
    t0 = time.perf_counter()
    for _ in range(draws):
        rho = rng.normal(size=(D, nchains))
        np.multiply(lam, theta, out=g); np.negative(g, out=g)
        lp0 = -0.5 * np.einsum("dc,dc->c", theta, lam * theta) - 0.5 * np.einsum("dc,dc->c", rho, m * rho)
        np.copyto(prop, theta)
        np.multiply(m, g, out=t); np.multiply(t, -half, out=t); np.add(rho, t, out=r)
        for _ in range(L):
            np.multiply(m, g, out=t); np.multiply(t, eps, out=t); np.add(r, t, out=r)
            np.multiply(r, eps, out=t); np.add(prop, t, out=prop)
            np.multiply(lam, prop, out=g); np.negative(g, out=g)
        np.multiply(m, g, out=t); np.multiply(t, half, out=t); np.add(r, t, out=r)
        lp1 = -0.5 * np.einsum("dc,dc->c", prop, lam * prop) - 0.5 * np.einsum("dc,dc->c", r, m * r)
        acc = np.log(rng.uniform(size=nchains)) < lp1 - lp0
        theta[:, acc] = prop[:, acc]
    return nchains * draws * L, time.perf_counter() - t0


if __name__ == "__main__":
    import json
    import sys

    a = sys.argv[1:]
    if a[0] == "batched":
        steps, secs = run_batched(int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), int(a[6]))
    else:
        steps, secs = run_chains(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), int(a[6]))
    print(json.dumps({"steps": steps, "seconds": secs}))
