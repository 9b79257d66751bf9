"""After choosing the best triple by role assignment, do small base shifts of rho / grad help further?"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D, N = 65536, 1024, 8
SL = 1 << 17  # slack in doubles (1 MiB)
raw = [torch.zeros(D * C + SL, dtype=torch.float64, device=dev) for _ in range(N)]
def view(i, shift):  # shift in doubles, multiple of 2 (16 B alignment)
    return raw[i][shift:shift + D * C].view(D, C)
def t(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def kd(a, b, c):
    return t(lambda: ops.kick_drift(a, a, b, b, c, None, 0.01, False, 0.0, True, 0.01))
rnd = random.Random(5)
best = (1e9, None)
for trial in range(30):
    i, j, k = rnd.sample(range(N), 3)
    us = kd(view(i, 0), view(j, 0), view(k, 0))
    if us < best[0]: best = (us, (i, j, k))
print("best triple by role assignment:", best)
i, j, k = best[1]
res = []
for trial in range(40):
    s1 = rnd.randrange(0, SL // 2) * 2 if trial else 0
    s2 = rnd.randrange(0, SL // 2) * 2 if trial else 0
    us = kd(view(i, 0), view(j, s1), view(k, s2))
    res.append((us, s1 * 8, s2 * 8))
res.sort()
print("shifts (bytes) of rho, grad, fastest 5:", [(round(u, 1), a, b) for u, a, b in res[:5]])
print("slowest 3:", [(round(u, 1), a, b) for u, a, b in res[-3:]], " unshifted:", [round(u, 1) for u, a, b in res if a == 0 and b == 0])
