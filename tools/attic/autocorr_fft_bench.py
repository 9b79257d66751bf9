"""bk_autocorr_fft (long-chain autocorrelation, own Stockham passes) against the same formula through torch.fft
(rocFFT; here for comparison only): time per call and agreement.  usage: autocorr_fft_bench.py [N=20000] [C=4096]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
x = torch.randn((N, C), dtype=torch.float64, device=ops.device).cumsum(0) * 0.01 + torch.randn((N, C), dtype=torch.float64, device=ops.device)
out = torch.empty_like(x)
size = 1 << int(np.ceil(np.log2(2 * N - 1)))


def via_torch():
    xc = x - x.mean(dim=0)
    f = torch.fft.rfft(xc, n=size, dim=0)
    pw = f.real * f.real + f.imag * f.imag
    return torch.fft.irfft(pw, n=size, dim=0)[:N] / x.var(dim=0, unbiased=False) / N


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms_own = timed(lambda: ops.autocorr_fft(x, out))
ref = via_torch()
ms_torch = timed(via_torch)
passes = 2 * len([1 for _ in range(0, int(np.log2(size)), 4)])  # radix 16 (a last pass of 8 / 4 / 2)
print(json.dumps({"N": N, "C": C, "fft_size": size, "ms_bk_autocorr_fft": round(ms_own, 3), "ms_torch_fft_formula": round(ms_torch, 3),
                  "max_abs_diff": float((out - ref).abs().max()), "passes": passes,
                  "pass_traffic_GB": round(passes * size * ((C + 1) // 2) * 32 / 1e9, 2),
                  "TBps_on_pass_traffic": round(passes * size * ((C + 1) // 2) * 32 / ms_own / 1e9, 2)}))
