"""bk_ess / bk_autocorr over [N, C] series (LDS-staged kernels): time per call.  usage: ess_bench.py [N C ...]
BK_ESS_LANE_PER_CHAIN=1: the one-lane-per-chain kernels instead."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
args = [int(a) for a in sys.argv[1:]] or [1000, 65536, 1000, 32768, 200, 65536, 4000, 16384, 12000, 4096]
for N, C in zip(args[::2], args[1::2]):
    g = torch.Generator(device=ops.device)
    g.manual_seed(1)
    PAD = int(os.environ.get("PAD", 0))  # extra columns in the row pitch of the series
    x = torch.randn((N, C + PAD), dtype=torch.float64, device=ops.device, generator=g)[:, :C]
    for t in range(1, N):
        x[t] += 0.8 * x[t - 1]
    e = torch.empty(C, dtype=torch.float64, device=ops.device)
    ops.ess(x, 0, e, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.ess(x, 0, e, None)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"N": N, "C": C, "ms_ess": round(e0.elapsed_time(e1) / 5, 3), "ess_mean": float(e.mean()),
                      "Gsamples_per_s": round(N * C / (e0.elapsed_time(e1) / 5) / 1e6, 1)}))
