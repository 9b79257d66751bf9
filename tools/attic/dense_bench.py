"""Dense mass matrix on the fp64 matrix cores: Y = M @ X at config-5 shape (D=512), timing
and TFLOP/s vs the 78.6 TFLOP/s fp64 MFMA peak; also HMC steps/s with metric_dense."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
from bayes_kit_amd import _lib

ops = _lib.default_ops()
D = int(os.environ.get("D", 512))
C = int(os.environ.get("C", 65536))
reps = int(os.environ.get("REPS", 20))
f = dict(dtype=torch.float64, device=ops.device)
M = torch.randn((D, D), **f)
X = torch.randn((D, C), **f)
Y = torch.empty((D, C), **f)
for _ in range(3):
    ops.dense_metric_apply(M, X, Y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.dense_metric_apply(M, X, Y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 2.0 * D * D * C
out = {"kernel": "k_dense_apply (bk_dense_metric_apply)", "D": D, "C": C, "ms": ms, "tflops_fp64": flop / ms / 1e9,
       "peak_tflops_fp64_mfma": 78.6, "frac": flop / ms / 1e9 / 78.6}
if os.environ.get("HMC", "1") == "1":
    lam = torch.logspace(0, 2, D, dtype=torch.float64)
    A = torch.randn((D, D), dtype=torch.float64) * 0.01
    Md = torch.diag(1.0 / lam) + A @ A.T / float(lam.max())
    L = 16
    s = bk.HMCDiag(bk.DiagGaussian(lam), 0.1, L, chains=C, seed=5, metric_dense=Md)
    s.sample()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        s.sample()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["hmc_dense_steps_per_sec"] = C * L * n / el
    out["hmc_dense_accept"] = s.accept_rate()
print(json.dumps(out))
