import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path[:0]=[R, R+"/bayes-kit_amd"]
import numpy as np, torch
import bayes_kit_amd as bk
from tests.helpers import load_case
from tests.sampler_parity import product_model, build_sampler
ops = bk._lib.default_ops()
for name in ("drghmc_funnel11_k3", "drghmc_funnel101_cfg4"):
    case, z = load_case(name)
    N, C, D = z["draws"].shape
    s = build_sampler(case, product_model(case["model"], ops), ops, case["seed"], chains=C)
    rel, ab, lrel = [], [], []
    for n in range(N):
        th, lp = s.sample()
        th, lp = th.cpu().numpy(), lp.cpu().numpy()
        w = z["draws"][n]
        ab.append(np.abs(th - w).max())
        rel.append((np.abs(th - w) / np.maximum(np.abs(w), 1e-300)).max())
        lrel.append((np.abs(lp - z["logp"][n]) / np.maximum(np.abs(z["logp"][n]), 1e-300)).max())
    print(name, N, C, D)
    print(" max abs err per draw:", " ".join("%.1e" % v for v in ab))
    print(" max rel err per draw:", " ".join("%.1e" % v for v in rel))
    print(" max rel err logp    :", " ".join("%.1e" % v for v in lrel))
    print(" scale of theta: max |theta|", float(np.abs(z["draws"]).max()))
