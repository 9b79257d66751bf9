"""Config-3 workload with the gradient supplied by PyTorch autograd (TorchModel) instead of the
C-ABI built-in target: what the model-opaque path costs when the model is user PyTorch code."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
C, D, L = int(os.environ.get("C", 65536)), 1024, 64
dev = torch.device("cuda", 0)
lam = torch.logspace(0, 4, D, dtype=torch.float64, device=dev)
res = {}
for name, model in (("autograd", bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), D)),
                    ("analytic torch ops", None)):
    if model is None:
        class Analytic:
            batched = True
            def dims(self): return D
            def log_density(self, Th): return -0.5 * (Th * Th * lam).sum(dim=1)
            def log_density_gradient(self, Th):
                t = Th * lam
                return -0.5 * (Th * t).sum(dim=1), -t
        model = Analytic()
    s = bk.HMCDiag(model, 0.006, L, chains=C, seed=20241, metric_diag=torch.ones(D, dtype=torch.float64))
    s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
    s.sample(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n):
        s.sample()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    res[name] = {"ms_per_draw": 1e3 * el, "steps_per_sec": C * L / el, "accept": s.accept_rate()}
print(json.dumps(res))
