"""Can the next draw's generator (VALU-bound) and the Welford update of the draw's diagnostics (HBM-bound) share the chip?
Config-4 shape (32,768 x 101): each alone, one after the other on one stream, and at once on two streams (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
dev = ops.device
C, D = 32768, 101
dp = (D + 7) // 8 * 8
state = torch.zeros((_lib.RNG_WORDS, C), dtype=torch.int64, device=dev)
ops.rng_init_philox(state, 1234, 0)
zt = torch.empty((C, dp), dtype=torch.float64, device=dev)
th = torch.randn((D, C), dtype=torch.float64, device=dev)
mean, m2 = torch.zeros_like(th), torch.zeros_like(th)
s2 = torch.cuda.Stream()
def zig(): ops.normals_chain_major(_lib.RNG_PHILOX, state, zt, D)
def wel(): ops.welford_update(mean, m2, th, 5)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
def both_seq(): zig(); wel()
def both_par():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        wel()
        ev2 = torch.cuda.Event(); ev2.record()
    zig()
    torch.cuda.current_stream().wait_event(ev2)
print({"zig_us": round(timed(zig), 1), "welford_us": round(timed(wel), 1), "one_after_the_other_us": round(timed(both_seq), 1),
       "two_streams_us": round(timed(both_par), 1)})
