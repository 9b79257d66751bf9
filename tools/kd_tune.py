"""A/B timing of kick+drift kernel variants on the config-3 shape (tuning aid, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
C, D = int(os.environ.get("C", 65536)), 1024
f = dict(dtype=torch.float64, device=ops.device)
th, rho, g = (torch.randn((D, C), **f) for _ in range(3))
m = torch.ones(D, **f)
variants = [int(v) for v in (sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "6"])]
res = {v: [] for v in variants}
for rep in range(6):
    for v in variants:
        os.environ["BK_KD_VARIANT_LIVE"] = str(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.kick_drift(th, th, rho, rho, g, m, 1e-9, False, 0.0, True, 1e-9)
        e0.record()
        for _ in range(20):
            ops.kick_drift(th, th, rho, rho, g, m, 1e-9, False, 0.0, True, 1e-9)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            res[v].append(e0.elapsed_time(e1) / 20)
for v in variants:
    t = sum(res[v]) / len(res[v])
    print(f"variant {v}: {t*1e3:.1f} us  {40.0*D*C/t/1e6:.0f} GB/s  (min {min(res[v])*1e3:.1f})")
