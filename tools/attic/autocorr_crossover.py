import os, sys, json
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N, C in ((256, 4096), (512, 4096), (1024, 4096), (2048, 4096), (4096, 2048), (8192, 1024), (16000, 512)):
    x = torch.randn((N, C), dtype=torch.float64, device=ops.device).cumsum(0)
    a, b = torch.empty_like(x), torch.empty_like(x)
    md = timed(lambda: ops.autocorr(x, a)); mf = timed(lambda: ops.autocorr_fft(x, b))
    print(json.dumps({"N": N, "C": C, "ms_direct": round(md, 3), "ms_fft": round(mf, 3), "max_abs_diff": float((a - b).abs().max())}))
