"""k_traj_gauss_q (bk_hmc_draw_gaussian) alone at 65,536 x 1024: time against the number of leapfrog steps, which
separates the inner loop (40 fp64 instructions per step per 8 rows) from the per-chunk loads / stores / sums.
ISO=1: iso Gaussian (no per-row constants); ONLY_L=n: one trajectory length (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
dev = ops.device
f64 = dict(dtype=torch.float64, device=dev)
th, out = torch.randn((D, C), **f64) * 0.01, torch.empty((D, C), **f64)
dp = (D + 7) // 8 * 8
zt = torch.randn((C, dp), **f64)
rho = zt[:, :D].t().contiguous()
lam = None if os.environ.get("ISO") else torch.logspace(0, 4, D, **f64)  # ISO=1: no per-row constant (scalar loads)
part, k0, k1, lp = torch.empty(12 * C, **f64), torch.empty(C, **f64), torch.empty(C, **f64), torch.empty(C, **f64)
for use_zt in (True, False):
    res = []
    for L in ([int(os.environ["ONLY_L"])] * 2 if os.environ.get("ONLY_L") else (0, 16, 64, 256)):
        def run():
            ops.hmc_draw_gaussian(th, out, None if use_zt else rho, zt if use_zt else None, lam, None, 0.006, L, part, k0, k1, lp)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append((L, e0.elapsed_time(e1) / 5))
    if os.environ.get("ONLY_L"):
        print("momentum", "chain-major" if use_zt else "state layout", res)
        continue
    (l0, t0), (l1, t1) = res[1], res[3]
    b = (t1 - t0) / (l1 - l0)
    print("momentum", "chain-major" if use_zt else "state layout", " ".join(f"L={l}: {t*1e3:.0f} us" for l, t in res),
          f"| per step {b*1e3:.2f} us = {5.0*D*C/(b*1e-3)/1e12:.1f} TFLOP/s; L=64 minus 64 steps: {(res[2][1]-64*b)*1e3:.0f} us")
