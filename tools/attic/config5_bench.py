"""BASELINE.json config 5 pieces on one MI355X: logistic regression N=1e6, D=512 (synthetic,
torch seed 20243) -- gradient for C chains = two fp64 MFMA GEMMs; HMC with a dense metric;
a short likelihood-annealed SMC with a Langevin move.  No reference oracle (tolerance parity
in tests/); numbers are reported against the 78.6 TFLOP/s fp64 MFMA peak."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

N = int(os.environ.get("N", 1_000_000)); D = int(os.environ.get("D", 512)); C = int(os.environ.get("C", 2048))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(20243)
X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=g) / D ** 0.5
tstar = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
y = (torch.rand(N, dtype=torch.float64, device=dev, generator=g) < torch.sigmoid(X @ tstar)).to(torch.float64)
model = bk.LogisticRegression(X, y, prior_scale=1.0)
th = torch.randn((D, C), dtype=torch.float64, device=dev, generator=g) * 0.1
grad = torch.empty_like(th); lp = torch.empty(C, dtype=torch.float64, device=dev)
model.bk_eval(th, grad, lp); torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    model.bk_eval(th, grad, lp)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / reps
flop = 2 * 2.0 * N * D * C
out = {"workload": "logistic regression N=%d D=%d, %d chains" % (N, D, C), "gradient_ms": 1e3 * el,
       "gradient_evals_per_sec": C / el, "tflops_fp64": flop / el / 1e12, "peak_tflops_fp64_mfma": 78.6,
       "frac": flop / el / 1e12 / 78.6}
# check against torch's own fp64 matmul on a slice of chains
z = X @ th[:, :8]; r = y[:, None] - torch.sigmoid(z); gref = X.t() @ r - th[:, :8]
out["grad_max_rel_err_vs_torch"] = float(((grad[:, :8] - gref).abs().max() / gref.abs().max()).item())
if os.environ.get("HMC", "1") == "1":
    Md = torch.eye(D, dtype=torch.float64) * 4.0 / N * D  # ~ posterior covariance scale (Fisher ~ N/(4D) I)
    s = bk.HMCDiag(model, 0.3, 8, chains=C, seed=20243, metric_dense=Md, init=th.t().contiguous().cpu())
    s.sample(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 2
    for _ in range(n):
        s.sample()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["hmc_dense_steps_per_sec"] = C * 8 * n / el
    out["hmc_dense_accept"] = s.accept_rate()
print(json.dumps(out))

if os.environ.get("SMC", "1") == "1":
    # the whole of config 5 on one GPU: likelihood-annealed SMC, HMC moves with a dense metric
    M, T = C, int(os.environ.get("T", 8))
    init = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)  # prior draws
    Md = torch.eye(D, dtype=torch.float64) * (4.0 * D / N)
    smc = bk.TemperedLikelihoodSMC(model, M, T, init, bk.hmc_kernel(0.5, 4, metric_dense=Md), seed=20243)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smc.run()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    post = smc.thetas.mean(dim=0)
    corr = torch.corrcoef(torch.stack([post, tstar]))[0, 1].item()
    print(json.dumps({"annealed_smc": {"particles": M, "temperatures": T, "move": "HMC L=4, dense metric",
                                       "seconds": el, "corr_posterior_mean_vs_truth": corr,
                                       "final_step_ess": smc.last_ess}}))
