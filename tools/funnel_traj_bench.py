"""Timing of the single-launch delayed-rejection proposal on the funnel vs steps and lanes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops()
D, C = 101, 32768
f = dict(dtype=torch.float64, device=ops.device)
th, rho, g = torch.randn((D, C), **f) * 0.3, torch.randn((D, C), **f), torch.randn((D, C), **f)
tho, rhoo, go = torch.empty((D, C), **f), torch.empty((D, C), **f), torch.empty((D, C), **f)
lp, kin = torch.empty(C, **f), torch.empty(C, **f)
for n in (32768, 4096, 512):
    for steps in (1, 10, 40, 160):
        a = [t[:, :n] for t in (tho, rhoo, go)]
        ops.dr_proposal_funnel(th, rho, g, None, a[0], a[1], a[2], lp[:n], kin[:n], None, 0.01, steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.dr_proposal_funnel(th, rho, g, None, a[0], a[1], a[2], lp[:n], kin[:n], None, 0.01, steps)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"lanes {n:6d} steps {steps:4d}: {t*1e3:8.1f} us   {t*1e3/steps:6.2f} us/step   {n*steps/t/1e6:8.1f} G lane-steps/s")
