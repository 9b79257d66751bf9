"""Extreme shapes (a million chains x 8 dims ... 2 chains x 140,000 dims) through HMC / MALA / DRGHMC:
watched chains must be bit-identical to the oracle (also exercises grid-dimension limits)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np, torch
import bayes_kit_amd as bk
from oracle import models as om, samplers as osamp
for (C, D, L) in ((1_000_000, 8, 3), (3, 20000, 2), (131072, 300, 2), (65, 70000, 1), (2, 140000, 1)):
    for alg in ("hmc", "mala", "drghmc"):
        lam = np.linspace(1.0, 2.0, D)
        try:
            if alg == "hmc":
                s = bk.HMCDiag(bk.DiagGaussian(lam), 0.01, L, chains=C, seed=5, path="step")
                mk = lambda sd: osamp.HMCDiag(om.DiagGaussian(lam), 0.01, L, seed=sd)
            elif alg == "mala":
                s = bk.MALA(bk.DiagGaussian(lam), 1e-4, chains=C, seed=5)
                mk = lambda sd: osamp.MALA(om.DiagGaussian(lam), 1e-4, seed=sd)
            else:
                s = bk.DrGhmcDiag(bk.DiagGaussian(lam), 2, [0.02, 0.01], [L, 2 * L], 0.3, chains=C, seed=5)
                mk = lambda sd: osamp.DrGhmcDiag(om.DiagGaussian(lam), 2, [0.02, 0.01], [L, 2 * L], 0.3, seed=sd)
            watch = sorted({0, C // 2, C - 1})
            got = []
            for n in range(3):
                th, lp = s.sample()
                got.append(th[watch].cpu().numpy())
            ok = True
            for j, c in enumerate(watch):
                o = mk(np.random.Philox(key=[5, c]))
                for n in range(3):
                    oth, _ = o.sample()
                    ok &= np.array_equal(oth, got[n][j])
            print(C, D, alg, "bit-identical" if ok else "MISMATCH", flush=True)
        except Exception as e:
            print(C, D, alg, "ERROR", repr(e)[:200], flush=True)
        del s
        torch.cuda.empty_cache()
