"""Config 3 (ill-conditioned Gaussian, lam = logspace(0, 4, 1024), eps = 0.006): ESS per second of the slowest and the fastest
coordinate against the trajectory length L -- what a config-3 USER should pick, next to the bench's fixed L = 64 (README).
    [C=16384] [DRAWS=100] python tools/cfg3_trajectory_length.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

C, D, N, eps = int(os.environ.get("C", 16384)), 1024, int(os.environ.get("DRAWS", 100)), 0.006
lam = torch.logspace(0, 4, D, dtype=torch.float64)
out = []
for L in (64, 128, 256, 384):
    s = bk.HMCDiag(bk.DiagGaussian(lam), eps, L, chains=C, seed=20241, metric_diag=torch.ones(D, dtype=torch.float64))
    s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(s._theta_dc.device)[:, None])
    for _ in range(5):
        s.sample()
    series = torch.empty((3, N, C), dtype=torch.float64, device=s._theta_dc.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in range(N):
        th, lp = s.sample()
        series[0, n], series[1, n], series[2, n] = th[:, 0], th[:, D - 1], lp
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ess = torch.stack([bk.ess(series[i]) for i in range(3)])
    ess = torch.where(ess > 0, ess, torch.full_like(ess, float(N))).clamp(max=float(N)).mean(dim=1)
    out.append({"L": L, "eps_times_L": eps * L, "ms_per_draw": 1e3 * el / N, "accept_rate": s.accept_rate(),
                "mean_ess_per_chain_of_%d_draws" % N: {"theta[0] (lam=1)": float(ess[0]), "theta[D-1] (lam=1e4)": float(ess[1]), "logp": float(ess[2])},
                "ess_per_sec_theta0_all_chains": float(ess[0]) * C / el, "leapfrog_steps_per_sec": C * L * N / el})
    del s, series
    torch.cuda.empty_cache()
print(json.dumps({"chains": C, "draws": N, "eps": eps, "by_L": out}, indent=1))
