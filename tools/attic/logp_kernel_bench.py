"""bk_target_diag_gaussian_grad WITH the log density (k_gauss_logp_v2: per-chain sums over D in the canonical quarter order) at
config-3 size: microseconds per launch, alone.  The gradient-only streaming kernel beside it for scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
ops = bk._lib.default_ops()
C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
pad = int(os.environ.get("PAD", 0))
f64 = dict(dtype=torch.float64, device=ops.device)
th = torch.randn((D, C + pad), **f64)[:, :C]
g = torch.empty((D, C + pad), **f64)[:, :C]
lp = torch.empty(C, **f64)
lam = torch.logspace(0, 4, D, **f64)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
a = t(lambda: ops.target_grad("diag_gaussian", lam, th, g, lp))
b = t(lambda: ops.target_grad("diag_gaussian", lam, th, g, None))
c = t(lambda: ops.target_grad("diag_gaussian", lam, th, None, lp))
print({"C": C, "D": D, "pad": pad, "grad+logp_us": round(a, 1), "TBps": round(16.0 * D * C / a / 1e6, 2), "grad_only_us": round(b, 1),
       "logp_only_us": round(c, 1), "logp_only_TBps": round(8.0 * D * C / c / 1e6, 2)})
