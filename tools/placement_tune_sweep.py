"""How good a role assignment does HMCDiag._tune_placement find as spares / trials grow? (config 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
lam = torch.logspace(0, 4, 1024, dtype=torch.float64)
for spares, trials in ((3, 12), (6, 30), (10, 60), (3, 60)):
    bk.HMCDiag.TUNE_PLACEMENT_SPARES, bk.HMCDiag.TUNE_PLACEMENT_TRIALS = spares, trials
    for rep in range(2):
        s = bk.HMCDiag(bk.DiagGaussian(lam), 0.006, 64, chains=65536, seed=1, fuse_builtin=False)
        print(spares, trials, {k: round(v, 4) for k, v in s.placement.items()}, flush=True)
        del s
        torch.cuda.empty_cache()
