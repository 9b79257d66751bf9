"""Does the integer-bound RNG kernel overlap with an HBM-bound kernel on a second stream?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
from bayes_kit_amd._engine import make_streams
ops = _lib.default_ops()
C, D = 65536, 1024
kind, st = make_streams(1, C, 0, False, ops.device)
zt = torch.empty((C, D), dtype=torch.float64, device=ops.device)
z = torch.empty((D, C), dtype=torch.float64, device=ops.device)
a = torch.randn((D, C), dtype=torch.float64, device=ops.device)
g = torch.randn((D, C), dtype=torch.float64, device=ops.device)
b = torch.empty_like(a)
fwd = torch.empty(C, dtype=torch.float64, device=ops.device)
rev = torch.empty(C, dtype=torch.float64, device=ops.device)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def mem(n):
    for _ in range(n):
        ops.mala_logq(a, g, b, g, 0.01, fwd, rev)

def rng_wave(n):
    for _ in range(n):
        ops.normals_chain_major(kind, st, zt, D)

def rng_lane(n):
    for _ in range(n):
        ops.momentum_refresh(kind, st, None, 0.0, 1.0, z, None, None)

def timeit(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

def both(rng, n_mem, n_rng):
    def run():
        with torch.cuda.stream(side):
            ops_s = ops
            rng(n_rng)
        mem(n_mem)
    return run

mem(3); rng_wave(2); rng_lane(2)
for name, rng, k in (("wave/chain", rng_wave, 20), ("lane/chain", rng_lane, 10)):
    t_mem = timeit(lambda: mem(40))
    t_rng = timeit(lambda: rng(k))
    side.wait_stream(main)
    t_both = timeit(both(rng, 40, k))
    print(f"{name}: mem x40 {t_mem:.2f} ms, rng x{k} {t_rng:.2f} ms, concurrently {t_both:.2f} ms "
          f"(sum {t_mem + t_rng:.2f}, max {max(t_mem, t_rng):.2f})")
