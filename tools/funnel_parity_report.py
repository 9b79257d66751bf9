"""Measured funnel parity error per draw: HIP against (a) the reference's goldens (np.dot order) and (b) the oracle summing
in the library's canonical order (oracle.models.FunnelCanonical: the two then differ inside exp() only), in units of the
flat SURVEY 8c bar (rel 1e-9 / abs 5e-11).  Also (c): how many exp() results differ between the device and NumPy."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/bayes-kit_amd"]
import numpy as np, torch
import bayes_kit_amd as bk
from oracle import models as om
from tests.helpers import load_case, oracle_sampler
from tests.sampler_parity import product_model, build_sampler
ops = bk._lib.default_ops()
bar = lambda got, want: float((np.abs(got - want) / (5e-11 + 1e-9 * np.abs(want))).max())
for name in ("drghmc_funnel11_k3", "drghmc_funnel101_cfg4", "drghmc_funnel17_k4", "drghmc_funnel129_k3"):
    case, z = load_case(name)
    N, C, D = z["draws"].shape
    s = build_sampler(case, product_model(case["model"], ops), ops, case["seed"], chains=C)
    orc = [oracle_sampler(case, c, model=om.FunnelCanonical(D)) for c in range(C)]
    eg, ec = [], []
    for n in range(N):
        th, lp = s.sample()
        th = th.cpu().numpy()
        oth = np.stack([o.sample()[0] for o in orc])
        eg.append(bar(th, z["draws"][n]))
        ec.append(bar(th, oth))
    print(name, N, C, D)
    print(" vs golden (np.dot order), per draw :", " ".join("%.1e" % v for v in eg))
    print(" vs canonical-order oracle, per draw:", " ".join("%.1e" % v for v in ec))
# (c) exp on the device against numpy
x = -torch.linspace(-12.0, 12.0, 2_000_001, dtype=torch.float64)
d = torch.exp(x.cuda()).cpu().numpy()
h = np.exp(x.numpy())
print("torch.exp on the device vs numpy.exp: %d of %d differ (all by one ulp: %s)" % ((d != h).sum(), d.size,
      bool(np.all(np.abs(d.view(np.int64) - h.view(np.int64)) <= 1))))
