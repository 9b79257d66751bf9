"""bk_gemm_chains (k_dense_apply) over shapes: time, TFLOP/s and the time lost against the in-loop rate of long K.
usage: gemm_k_sweep.py [shapes "R,K,C;R,K,C;..."]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
f = dict(dtype=torch.float64, device=ops.device)
default = ";".join(f"{R},{K},65536" for R in (512, 1024) for K in (256, 512, 1024, 2048, 4096))
shapes = [tuple(int(v) for v in s.split(",")) for s in (sys.argv[1] if len(sys.argv) > 1 else default).split(";")]
PAD = int(os.environ.get("PAD", 0))  # extra columns in the row pitch of X and Y
for R, K, C in shapes:
    A = torch.randn((R, K), **f); X = torch.randn((K, C + PAD), **f)[:, :C]; Y = torch.empty((R, C + PAD), **f)[:, :C]
    for _ in range(2): ops.gemm_chains(A, X, Y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_chains(A, X, Y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flop = 2.0 * R * K * C
    print(json.dumps({"pad": PAD, "R": R, "K": K, "C": C, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 2),
                      "tiles_per_slot": (R // 128) * (C // 128) / 512, "lost_us_vs_68.6": round(1e3 * (ms - flop / 68.6e9), 1)}))
    del A, X, Y
