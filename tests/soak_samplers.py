"""Randomised differential soak: many-chain HIP samplers vs the NumPy oracle (one object per chain).

Random algorithm (HMC / MALA / DRGHMC / Metropolis), target (iso / diag Gaussian), dims (1..70: both
generator kernels), chains (odd and even: scalar and 16-byte kernels), step sizes, trajectory lengths,
metric on/off, fused / step-by-step, hipGraph on/off, RNG prefetch on/off.  For a few watched chains
theta must be bit-identical to the oracle at every draw and the stream state equal at the end.
SECONDS env var = duration (default 60); SEED = rng seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
import bayes_kit_amd as bk
from bayes_kit_amd.metropolis import ChainRng
from oracle import models as om
from oracle import samplers as osamp


def one(rng, it):
    alg = rng.choice(["hmc", "mala", "drghmc", "metropolis"])
    D = int(rng.choice([1, 2, 7, 31, 32, 33, 40, 64, 70]))
    C = int(rng.choice([1, 2, 3, 63, 64, 65, 130, 257, 500]))
    N = int(rng.integers(3, 12))
    seed = int(rng.integers(1, 2**40))
    iso = rng.random() < 0.4
    lam = None if iso else np.logspace(0, rng.uniform(0.2, 1.0), D)
    tgt = bk.IsoGaussian(D) if iso else bk.DiagGaussian(lam)
    otgt = (lambda: om.IsoGaussian(D)) if iso else (lambda: om.DiagGaussian(lam))
    metric = np.linspace(0.8, 1.2, D) if (rng.random() < 0.4 and alg in ("hmc", "drghmc")) else None
    eps = float(rng.uniform(0.02, 0.3))
    desc = dict(alg=alg, D=D, C=C, N=N, seed=seed, iso=iso, metric=metric is not None, eps=eps)
    if alg == "hmc":
        L = int(rng.integers(0, 7))
        kw = dict(fuse_builtin=bool(rng.integers(0, 2)), graph=bool(rng.integers(0, 2)),
                  prefetch_rng=bool(rng.integers(0, 2)))
        desc.update(L=L, **kw)
        s = bk.HMCDiag(tgt, eps, L, metric_diag=metric, chains=C, seed=seed, **kw)
        mk = lambda sd: osamp.HMCDiag(otgt(), eps, L, metric_diag=metric, seed=sd)
    elif alg == "mala":
        kw = dict(graph=bool(rng.integers(0, 2)), prefetch_rng=bool(rng.integers(0, 2)))
        desc.update(**kw)
        s = bk.MALA(tgt, eps * 0.3, chains=C, seed=seed, **kw)
        mk = lambda sd: osamp.MALA(otgt(), eps * 0.3, seed=sd)
    elif alg == "drghmc":
        K = int(rng.integers(1, 4))
        sizes = [float(eps * 2.0 / (2 ** k)) for k in range(K)]
        counts = [int(rng.integers(1, 4)) * (2 ** k) for k in range(K)]
        damp = float(rng.uniform(0.05, 1.0))
        pr = bool(rng.integers(0, 2))
        desc.update(K=K, sizes=sizes, counts=counts, damp=damp, prob_retry=pr)
        s = bk.DrGhmcDiag(tgt, K, sizes, counts, damp, metric_diag=metric, chains=C, seed=seed, prob_retry=pr)
        mk = lambda sd: osamp.DrGhmcDiag(otgt(), K, sizes, counts, damp, metric_diag=metric, seed=sd, prob_retry=pr)
    else:
        scale = float(rng.uniform(0.2, 1.5))
        pseed = seed + 17
        prop = ChainRng(pseed, C)
        desc.update(scale=scale)
        s = bk.Metropolis(tgt, lambda Th: prop.normal(Th, scale), chains=C, seed=seed)

        def mk(sd, pseed=pseed, scale=scale):
            c = int(sd.state["state"]["key"][1])
            g = np.random.Generator(np.random.Philox(key=[pseed, c]))
            return osamp.Metropolis(otgt(), lambda th: g.normal(loc=th, scale=scale), seed=sd)
    watch = sorted(set([0, C // 2, C - 1]))
    got = []
    for n in range(N):
        th, _ = s.sample()
        got.append(th[watch].cpu().numpy())
    state = s.rng_state()
    for j, c in enumerate(watch):
        o = mk(np.random.Philox(key=[seed, c]))
        for n in range(N):
            oth, _ = o.sample()
            assert np.array_equal(oth, got[n][j]), ("theta", desc, c, n)
        st = o._rng.bit_generator.state
        assert [int(v) for v in st["state"]["counter"]] == [int(v) for v in state[2:6, c]], ("counter", desc, c)
        assert int(st["buffer_pos"]) == int(state[10, c]), ("pos", desc, c)
    return alg


if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
    budget = float(os.environ.get("SECONDS", 60))
    t0, counts = time.time(), {}
    while time.time() - t0 < budget:
        a = one(rng, sum(counts.values()))
        counts[a] = counts.get(a, 0) + 1
    print("soak ok:", sum(counts.values()), "random sampler configurations", counts, f"in {time.time()-t0:.0f} s")
