"""MALA many-chain throughput at config-3 shape (88*D algorithmic bytes per chain-draw)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
lam = torch.logspace(0, 4, D, dtype=torch.float64)
s = bk.MALA(bk.DiagGaussian(lam), 5e-5, chains=C, seed=7, graph=os.environ.get("GRAPH", "0") == "1",
            prefetch_rng={"0": False, "1": True}.get(os.environ.get("PREFETCH", ""), None),
            two_pass={"0": False, "1": True}.get(os.environ.get("TWO_PASS", ""), None),
            path="auto" if os.environ.get("INLINED", "1") == "1" else "opaque")  # INLINED=0: the model-opaque pair {gradient op, bk_mala_step}
for k in ("serialize_step", "generate_with", "generator_workgroups"):   # e.g. serialize_step=0 generate_with=step
    if os.environ.get(k):
        setattr(s, k, os.environ[k] if k == "generate_with" else int(os.environ[k]))
s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(s._theta_dc.device)[:, None])
s.refresh_cache()
for _ in range(3):
    s.sample()
torch.cuda.synchronize(); n = 20
t0 = time.perf_counter()
for _ in range(n):
    s.sample()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
print(json.dumps({"workload": f"MALA diag Gaussian D={D} x {C} chains", "ms_per_draw": 1e3 * el, "draws_per_sec": C / el,
                  "achieved_GBps_88D_model": 88.0 * D * C / el / 1e9, "frac_of_8TBps": 88.0 * D * C / el / 8e12,
                  "accept_rate": s.accept_rate(), "path": s.path, "prefetch": s._prefetch}))
