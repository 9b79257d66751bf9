"""bk_dr_proposal_funnel alone (HIP events): one whole delayed-rejection proposal on Neal's funnel for n chains of
a 32,768-chain parent set, at the (lanes, steps) of config 4's seven trajectories.  `dev=1` passes the lane
count in device memory (launch sized for the parent set, 16 lanes per chain), `dev=0` on the host.
    python tools/funnel_traj_bench.py            # BK_FUNNEL_GEOMETRY=wide|narrow to force a geometry"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

if os.environ.get("BK_LIB"):  # A/B against another build of the library (e.g. tools/kbench/bin/libbkhip_r2_targets.so)
    _lib._LIB_PATH = os.path.abspath(os.environ["BK_LIB"])
ops = _lib.default_ops()
dev = ops.device
C, D = 32768, int(os.environ.get("D", 101))
PAD = int(os.environ.get("PAD", 0))  # extra columns per row: leading dimension C + PAD (row stride not a power of two)
f64 = dict(dtype=torch.float64, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
def alloc():
    return torch.zeros((D, C + PAD), **f64)[:, :C]
th = alloc(); th.copy_(torch.randn((D, C), generator=g, **f64)); th[0] *= 3.0
th[1:] *= torch.exp(0.5 * th[0])
rho = alloc(); rho.copy_(torch.randn((D, C), generator=g, **f64))
grad, lp = alloc(), torch.empty(C, **f64)
ops.target_grad("funnel", None, th, grad, lp)
out = [alloc() for _ in range(3)]
lpo, kin = torch.empty(C, **f64), torch.empty(C, **f64)
idx_all = torch.randperm(C, device=dev, generator=g).to(torch.int32)
res = []
for tag, n, h, steps in [("P0", 32768, 0.2, 10), ("P1", 4091, 0.05, 40), ("G0(P1)", 4091, 0.2, 10), ("P2", 2330, 0.0125, 160),
                         ("G0(P2)", 2330, 0.2, 10), ("G1(P2)", 780, 0.05, 40), ("G0(G1(P2))", 780, 0.2, 10)]:
    idx = None if n == C else torch.sort(idx_all[:n]).values.contiguous()
    n_dev = None if n == C else torch.tensor([n], dtype=torch.int32, device=dev)
    def timed(st):
        def run():
            ops.dr_proposal_funnel(th, rho, grad, idx, out[0], out[1], out[2], lpo, kin, None, h, st, n_dev=n_dev)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    us, us1 = timed(steps), timed(1)
    res.append({"traj": tag, "lanes": n, "steps": steps, "us": round(us, 2), "us_one_step": round(us1, 2),
                "us_per_extra_step": round((us - us1) / (steps - 1), 3),
                "gflops_13D": round(n * steps * 13.0 * D / us / 1e3, 1)})
print(json.dumps({"D": D, "pad": PAD, "lib": os.environ.get("BK_LIB"), "geometry_env": os.environ.get("BK_FUNNEL_GEOMETRY"), "trajectories": res,
                  "sum_us": round(sum(r["us"] for r in res), 1)}))
