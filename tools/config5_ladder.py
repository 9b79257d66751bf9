"""BASELINE.json configs[4] end to end: the whole annealed ladder on the logistic regression (N = 1e6, D = 512, 2,048
particles), adaptive temperatures (ESS target), HMC moves with the metric adapted to the particles.
usage: [ESS=0.5] [EPS=0.35] [L=3] [N=1000000] [D=512] [C=2048] python tools/config5_ladder.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

N, D, C = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 512)), int(os.environ.get("C", 2048))
target, eps, L = float(os.environ.get("ESS", 0.5)), float(os.environ.get("EPS", 0.35)), int(os.environ.get("L", 3))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(20243)
X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=g) / D ** 0.5
tstar = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
y = (torch.rand(N, dtype=torch.float64, device=dev, generator=g) < torch.sigmoid(X @ tstar)).to(torch.float64)
model = bk.LogisticRegression(X, y, prior_scale=1.0)
init = torch.randn((C, D), dtype=torch.float64, device=dev, generator=g)
smc = bk.TemperedLikelihoodSMC(model, C, 1, init, bk.hmc_kernel(eps, L, adapt_metric=True), seed=20243, adaptive=target)
torch.cuda.synchronize()
t0 = time.perf_counter()
smc.run()
torch.cuda.synchronize()
el = time.perf_counter() - t0
post = smc.thetas.mean(dim=0)
T = smc.temperatures
evals = len(T) * L + 1  # model evaluations of all particles: L per move (+ the first temperature's; then bk_retemper)
print(json.dumps({"N": N, "D": D, "particles": C, "ess_target": target, "eps": eps, "L": L, "temperatures": len(T),
                  "first_temperatures": T[:4], "min_ess": min(smc.ess_history), "seconds": el,
                  "tflops_fp64": evals * 4.0 * N * D * C / el / 1e12,
                  "accept_min_mean": [min(smc.kernel.accept_rates), sum(smc.kernel.accept_rates) / len(T)],
                  "corr_posterior_mean_vs_truth": float(torch.corrcoef(torch.stack([post, tstar]))[0, 1]),
                  "rel_err_vs_truth": float(((post - tstar).norm() / tstar.norm()).item()),
                  "unique_particles": int(torch.unique(smc.thetas[:, 0]).numel())}))
