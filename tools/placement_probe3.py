"""Spread of the MALA kernels (4-5 arrays at equal offsets) over role assignments of a pool of allocations."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D, N = 65536, 1024, 10
arrs = [torch.zeros((D, C), dtype=torch.float64, device=dev) for _ in range(N)]
fwd = torch.empty(C, dtype=torch.float64, device=dev); rev = torch.empty_like(fwd)
mask = (torch.rand(C, device=dev) < 0.8).to(torch.uint8)
def t(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rnd = random.Random(3)
res = []
for trial in range(16):
    th, g, thp, gp, z, out = (arrs[i] for i in rnd.sample(range(N), 6))
    a = t(lambda: ops.mala_propose_from_normals(th, g, z, thp, 0.01, 0.1))
    b = t(lambda: ops.mala_logq(th, g, thp, gp, 0.01, fwd, rev))
    c = t(lambda: ops.select_columns(mask, th, thp, g, gp, out))
    res.append((a + b + c, a, b, c))
    print(f"trial {trial}: propose {a:.0f}  logq {b:.0f}  select+copy {c:.0f}  sum {a+b+c:.0f} us")
print("best", min(res)[0], "worst", max(res)[0])
