"""k_kick_drift_tr at config-3 size: the kick + drift with a ROW-MAJOR (C, D) gradient (what PyTorch user code
returns by default), turned through 64 x 64 LDS tiles inside the kernel; against the chain-contiguous kernel
on the same arrays.  40 D algorithmic bytes per chain-step either way."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
f64 = dict(dtype=torch.float64, device=ops.device)
PAD, GPAD = int(os.environ.get("PAD", 0)), int(os.environ.get("GPAD", 0))  # extra columns in the row pitch of theta / rho, of the gradient
th, rho = torch.randn((D, C + PAD), **f64)[:, :C], torch.randn((D, C + PAD), **f64)[:, :C]
g_rm = torch.randn((C, D + GPAD), **f64)[:, :D]   # row-major (C, D): dimension-contiguous
g_cc = g_rm.t().contiguous()               # the same values, chain-contiguous [D, C]
m = torch.ones(D, **f64)
out = {}
for name, g in (("row_major_gradient (k_kick_drift_tr)", g_rm.t()), ("chain_contiguous_gradient (k_kick_drift_v2)", g_cc)):
    a, b = torch.empty((D, C + PAD), **f64)[:, :C], torch.empty((D, C + PAD), **f64)[:, :C]
    a.copy_(th); b.copy_(rho)
    for _ in range(3):
        ops.kick_drift(a, a, b, b, g, m, 0.006, False, 0.0, True, 0.006)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.kick_drift(a, a, b, b, g, m, 0.006, False, 0.0, True, 0.006)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out[name] = {"ms": round(ms, 4), "TBps_40D_model": round(40.0 * D * C / ms / 1e9, 3), "frac_of_8TBps": round(40.0 * D * C / ms / 1e9 / 8, 3)}
    res = a.clone()
    out[name]["checksum"] = float(res[::97, ::991].sum().item())
assert list(out.values())[0]["checksum"] == list(out.values())[1]["checksum"]
out["pad_theta_rho"], out["pad_gradient"] = PAD, GPAD
print(json.dumps(out))
