"""kick+drift bandwidth by gradient layout: chain-contiguous (streamed) vs dimension-contiguous
(a row-major (C, D) model output, transposed through LDS tiles inside the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
C, D = 65536, 1024
f = dict(dtype=torch.float64, device=ops.device)
th, rho = torch.randn((D, C), **f), torch.randn((D, C), **f)
g_chain = torch.randn((D, C), **f)
g_row = torch.randn((C, D), **f).t()   # logical [D, C], strides (1, D)
m = torch.ones(D, **f)
for name, g in (("chain-contiguous grad", g_chain), ("row-major (C,D) grad via LDS tiles", g_row)):
    for _ in range(2):
        ops.kick_drift(th, th, rho, rho, g, m, 1e-9, False, 0.0, True, 1e-9)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.kick_drift(th, th, rho, rho, g, m, 1e-9, False, 0.0, True, 1e-9)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{name}: {t*1e3:.1f} us  {40.0*D*C/t/1e6:.0f} GB/s")
