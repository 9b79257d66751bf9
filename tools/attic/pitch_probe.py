"""Row pitch of the state arrays vs kernel time at config-3 size ([1024 x 65,536] fp64): the streaming kick + drift, the
column-walking reductions (finish / kinetic energy, Gaussian gradient + log density) and the select / blend, with the rows
at a 512-KiB pitch and PAD columns further apart.  usage: pitch_probe.py [PAD ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
D, C = 1024, 65536
f64 = dict(dtype=torch.float64, device=ops.device)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / reps, 1)


for pad in [int(a) for a in sys.argv[1:]] or [0, 72]:
    mk = lambda: torch.randn((D, C + pad), **f64)[:, :C]
    th, thp, rho, g, gp, out = mk(), mk(), mk(), mk(), mk(), mk()
    lam = torch.linspace(1.0, 2.0, D, **f64)
    kin, lp = torch.empty(C, **f64), torch.empty(C, **f64)
    mask = (torch.rand(C, device=ops.device) < 0.7).to(torch.uint8)
    r = {"pad": pad}
    r["kick_drift_us"] = timed(lambda: ops.kick_drift(thp, thp, rho, rho, g, None, 0.01, False, 0.0, True, 0.01))
    r["gauss_grad_us"] = timed(lambda: ops.target_grad("diag_gaussian", lam, th, g, None))
    r["gauss_grad_logp_us"] = timed(lambda: ops.target_grad("diag_gaussian", lam, th, g, lp))
    r["finish_us"] = timed(lambda: ops.leapfrog_finish(rho, None, g, None, 0.005, False, kin))
    r["blend_us"] = timed(lambda: ops.blend_columns(mask, th, thp, out))
    r["select_us"] = timed(lambda: ops.select_columns(mask, g, gp))
    print(json.dumps(r))
    del th, thp, rho, g, gp, out
