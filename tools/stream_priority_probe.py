"""Does running the sampler on a HIGH-priority stream (the RNG side stream stays at normal priority)
reduce the cost of the concurrent generator?  Config 3, same sampler objects, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
lam = torch.logspace(0, 4, 1024, dtype=torch.float64)
def run(stream, prefetch):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        s = bk.HMCDiag(bk.DiagGaussian(lam), 0.006, 64, chains=65536, seed=20241, fuse_builtin=False,
                       metric_diag=torch.ones(1024, dtype=torch.float64), prefetch_rng=prefetch)
        s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(s._theta_dc.device)[:, None])
        for _ in range(3): s.sample()
        names = ("bk_leapfrog_kick_drift", "bk_target_diag_gaussian_grad", "bk_select_columns", "bk_leapfrog_finish")
        s._ops.timed = {n: [] for n in names}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): s.sample()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 10
        timed, s._ops.timed = s._ops.timed, None
        per = {n.replace("bk_", ""): round(sum(a.elapsed_time(b) for a, b in v) / 10, 2) for n, v in timed.items()}
        return ms, s.placement.get("step_ms_chosen", s.placement.get("kick_drift_ms_chosen")), per
hi = torch.cuda.Stream(priority=-1)
for rep in range(int(os.environ.get("REPS", 2))):
    for name, st, pf in (("default stream, prefetch", None, True), ("high-priority stream, prefetch", hi, True),
                         ("default stream, inline rng", None, False)):
        ms, kd, per = run(st, pf)
        print(f"{name}: {ms:.2f} ms per draw (tuned kick+drift {kd*1e3:.1f} us) per-draw kernel ms {per}", flush=True)
