"""Is the streaming rate a property of each allocation (then the best ones can be picked one by one)?
Times an in-place one-array kernel on N candidate arrays, then kick+drift on the 3 best / 3 worst."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D, N = 65536, 1024, int(os.environ.get("N", 12))
arrs = [torch.zeros((D, C), dtype=torch.float64, device=dev) for _ in range(N)]
def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
single = []
for i, a in enumerate(arrs):
    us = t(lambda: ops.target_grad("iso_gaussian", None, a, a, None))
    single.append((us, i))
    print(f"array {i}: in-place pass {us:.1f} us ({16*D*C/us/1e6:.2f} TB/s)  ptr {a.data_ptr():#x}")
single.sort()
def kd(idx):
    th, rho, g = (arrs[i] for i in idx)
    return t(lambda: ops.kick_drift(th, th, rho, rho, g, None, 0.01, False, 0.0, True, 0.01))
best = [i for _, i in single[:3]]; worst = [i for _, i in single[-3:]]; mid = [i for _, i in single[4:7]]
print("kick+drift on the 3 best arrays :", best, f"{kd(best):.1f} us")
print("kick+drift on 3 middle arrays   :", mid, f"{kd(mid):.1f} us")
print("kick+drift on the 3 worst arrays:", worst, f"{kd(worst):.1f} us")
import itertools, random
random.seed(1)
combos = random.sample(list(itertools.combinations(range(N), 3)), 12)
for c in combos:
    print("combo", c, f"{kd(c):.1f} us", " sum of single-array times", f"{sum(dict((i, u) for u, i in single)[i] for i in c):.1f}")
