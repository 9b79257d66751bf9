"""Does the placement of an allocation batch change the streaming rate?  (torch allocations)
Allocates K groups of (theta, rho, grad) [D, C] tensors and times bk_leapfrog_kick_drift and the
gradient op on each group."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D, K = 65536, 1024, int(os.environ.get("K", 8))
lam = torch.logspace(0, 4, D, dtype=torch.float64, device=dev)
groups = []
for k in range(K):
    g = [torch.zeros((D, C), dtype=torch.float64, device=dev) for _ in range(3)]
    groups.append(g)
def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for k, (th, rho, g) in enumerate(groups):
    kd = t(lambda: ops.kick_drift(th, th, rho, rho, g, None, 0.01, False, 0.0, True, 0.01))
    gr = t(lambda: ops.target_grad("diag_gaussian", lam, th, g, None))
    print(f"group {k}: kick+drift {kd:.1f} us ({40*D*C/kd/1e6:.2f} TB/s)  gradient {gr:.1f} us ({16*D*C/gr/1e6:.2f} TB/s)  step {kd+gr:.1f} us")
