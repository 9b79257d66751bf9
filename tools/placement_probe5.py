"""Is the fast mode (kick+drift ~0.412 ms) available in every process if the pool is large enough?
30 arrays of 512 MiB, 300 random triples; prints the fastest few and the fastest triple among the first
8 arrays only (what a small pool sees)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D, N = 65536, 1024, int(os.environ.get("N", 30))
arrs = [torch.zeros((D, C), dtype=torch.float64, device=dev) for _ in range(N)]
def kd(i, j, k, n=6):
    a, b, c = arrs[i], arrs[j], arrs[k]
    f = lambda: ops.kick_drift(a, a, b, b, c, None, 0.01, False, 0.0, True, 0.01)
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rnd = random.Random(int(os.environ.get("SEED", 1)))
small = sorted((kd(*t), t) for t in [tuple(rnd.sample(range(8), 3)) for _ in range(40)])
big = sorted((kd(*t), t) for t in [tuple(rnd.sample(range(N), 3)) for _ in range(300)])
print("pool of 8 : best", [(round(u, 1), t) for u, t in small[:2]])
print(f"pool of {N}: best", [(round(u, 1), t) for u, t in big[:4]], " median", round(big[len(big)//2][0], 1), " worst", round(big[-1][0], 1))
