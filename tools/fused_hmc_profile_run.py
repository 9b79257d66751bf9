"""Built-in config-3 target, whole-draw HMC path (bk_hmc_draw_gaussian), for rocprofv3: warm-up, then N draws."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
import bayes_kit_amd as bk
C, D, N = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024)), int(os.environ.get("N", 20))
metric = np.ones(D) if os.environ.get("METRIC", "1") == "1" else None  # (BASELINE config 3: diag metric of ones)
s = bk.HMCDiag(bk.DiagGaussian(np.logspace(0, 4, D)), 0.006, 64, metric_diag=metric, chains=C, seed=1, path="auto",
               prefetch_rng={"0": False, "1": True}.get(os.environ.get("PREFETCH", ""), None))
for _ in range(5):
    s.sample()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    s.sample()
torch.cuda.synchronize()
print({"ms_per_draw": 1e3 * (time.perf_counter() - t0) / N, "prefetch": s._prefetch, "fused_zt": s._fused_zt})
