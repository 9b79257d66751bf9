"""Launch-bound regime (config-2 shape: 4096 chains x D=128, L=32) with a USER PyTorch model:
eager launches vs hipGraph replay of the whole draw (graph=True), autograd and analytic torch ops."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
C, D, L = int(os.environ.get("C", 4096)), int(os.environ.get("D", 128)), 32
dev = torch.device("cuda", 0)


class Analytic:
    batched = True
    def dims(self): return D
    def log_density(self, Th): return -0.5 * (Th * Th).sum(dim=1)
    def log_density_gradient(self, Th): return -0.5 * (Th * Th).sum(dim=1), -Th


res = {}
for name, mk in (("autograd", lambda: bk.TorchModel(lambda Th: -0.5 * (Th * Th).sum(dim=1), D)),
                 ("analytic torch ops", Analytic), ("built-in target, opaque path", lambda: bk.IsoGaussian(D))):
    for graph in (False, True):
        try:
            s = bk.HMCDiag(mk(), 0.05, L, chains=C, seed=20240, graph=graph, fuse_builtin=False)
            for _ in range(3):
                s.sample()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 20
            for _ in range(n):
                s.sample()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n
            res[f"{name}, graph={graph}"] = {"ms_per_draw": round(1e3 * el, 4), "steps_per_sec": C * L / el,
                                             "accept": s.accept_rate()}
        except Exception as e:  # report, do not hide
            res[f"{name}, graph={graph}"] = {"error": repr(e)[:200]}
print(json.dumps(res, indent=1))
