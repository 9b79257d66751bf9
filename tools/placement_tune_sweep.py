"""How good a role assignment does HMCDiag._tune_placement find as spares / trials grow? (config 3)
One sampler per process invocation: SPARES / TRIALS from the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
lam = torch.logspace(0, 4, 1024, dtype=torch.float64)
bk.HMCDiag.TUNE_PLACEMENT_SPARES = int(os.environ.get("SPARES", 3))
bk.HMCDiag.TUNE_PLACEMENT_TRIALS = int(os.environ.get("TRIALS", 30))
s = bk.HMCDiag(bk.DiagGaussian(lam), 0.006, 64, chains=65536, seed=1, fuse_builtin=False)
print(os.environ.get("SPARES", 3), os.environ.get("TRIALS", 30), {k: round(v, 4) for k, v in s.placement.items()}, flush=True)
