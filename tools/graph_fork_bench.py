"""hipGraph replay of a draw with the next draw's generator as a parallel branch (prefetch_rng=True)
vs one serial graph (prefetch_rng=False); config-2 shape, built-in target, same box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
res = {}
for C, D, L in ((4096, 128, 32), (1024, 64, 16), (16384, 128, 32)):
    for fused in (True, False):
        for pf in (False, True):
            s = bk.HMCDiag(bk.IsoGaussian(D), 0.05, L, chains=C, seed=1, graph=True, prefetch_rng=pf, fuse_builtin=fused)
            for _ in range(4):
                s.sample()
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
            for _ in range(n):
                s.sample()
            torch.cuda.synchronize()
            res[f"C={C} D={D} L={L} fused={fused} branch={pf}"] = round(1e6 * (time.perf_counter() - t0) / n, 1)
print(json.dumps(res, indent=1))
