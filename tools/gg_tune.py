"""A/B timing of Gaussian-gradient kernel variants (tuning aid): one process per variant."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
    import torch
    from bayes_kit_amd import _lib
    ops = _lib.default_ops()
    C, D = 65536, 1024
    f = dict(dtype=torch.float64, device=ops.device)
    th, g = torch.randn((D, C), **f), torch.empty((D, C), **f)
    lam = torch.ones(D, **f)
    best = []
    for rep in range(4):
        ops.target_grad("diag_gaussian", lam, th, g, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.target_grad("diag_gaussian", lam, th, g, None)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20)
    t = min(best)
    print(f"variant {os.environ.get('BK_GG_VARIANT')}: {t*1e3:.1f} us {16.0*D*C/t/1e6:.0f} GB/s")
else:
    for v in ["0", "1", "2", "3", "4", "0", "2"]:
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, BK_GG_VARIANT=v))
