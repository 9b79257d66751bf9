"""Diagnostics kernels at config-4 size (32,768 chains x 101 dims, 1000 draws of 4 tracked series), for
rocprofv3: Welford updates, R-hat from the moments, ESS of stored series, split / rank-normalised R-hat."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
C, D, N = int(os.environ.get("C", 32768)), int(os.environ.get("D", 101)), int(os.environ.get("N", 1000))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
mom = bk.RunningMoments(D, C)
th = torch.randn((D, C), dtype=torch.float64, device=dev, generator=g)
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
out = {"shape": {"chains": C, "dims": D, "draws": N}}
t = timed(lambda: mom.update(th), 50)
out["welford_update"] = {"us": 1e6 * t, "algorithmic_bytes": 40 * D * C, "GBps": 40 * D * C / t / 1e9}
t = timed(lambda: mom.rhat(), 5)
out["rhat_from_moments"] = {"us": 1e6 * t, "algorithmic_bytes": 2 * 16 * D * C, "note": "two passes over (mean, M2) + two tiny gathers + host read"}
# AR(1) series, phi = 0.9 (IAT ~ 19): draw-major [N, C]
x = torch.empty((N, C), dtype=torch.float64, device=dev)
e = torch.randn((N, C), dtype=torch.float64, device=dev, generator=g)
x[0] = e[0]
for i in range(1, N):
    x[i] = 0.9 * x[i - 1] + e[i]
t = timed(lambda: bk.ess(x), 5)
out["ess"] = {"us": 1e6 * t, "algorithmic_bytes": 8 * N * C, "GBps": 8 * N * C / t / 1e9, "chains_per_sec": C / t,
              "mean_ess": float(bk.ess(x).mean().item())}
t = timed(lambda: bk.rhat(x), 5)
out["rhat_of_series"] = {"us": 1e6 * t}
t = timed(lambda: bk.split_rhat(x), 3)
out["split_rhat_of_series"] = {"us": 1e6 * t}
xs = x[:, :2048].contiguous()
t = timed(lambda: bk.rank_normalized_rhat(xs), 3)
out["rank_normalized_rhat_2048_chains"] = {"us": 1e6 * t}
t = timed(lambda: bk.autocorr(xs), 2)
out["autocorr_all_lags_2048_chains"] = {"us": 1e6 * t}
print(json.dumps(out))
