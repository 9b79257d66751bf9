"""One leapfrog step {gradient, kick, drift} of a separable density as ONE streaming launch (bke::k_step) at config-3 shape:
microseconds per launch and the fraction of 8 TB/s its 32*D*C algorithmic bytes reach (measured with 1 / 2 / 4 rows per thread of the
non-temporal variant: 0.789 / 0.817 / 0.706; the library runs two)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayes-kit_amd"))
import bayes_kit_amd as bk

D, C = 1024, 65536
lam = torch.logspace(0, 2, D, dtype=torch.float64)
m = bk.DiagGaussian(lam.numpy())
th = torch.randn((D, C), dtype=torch.float64, device="cuda") * 0.1
rho = torch.randn((D, C), dtype=torch.float64, device="cuda")
metric = torch.ones(D, dtype=torch.float64, device="cuda")
for met in (metric, None):
    for _ in range(5):
        m.bk_leapfrog_step(th, rho, met, 1e-4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        m.bk_leapfrog_step(th, rho, met, 1e-4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("metric" if met is not None else "no metric", "us per step %.1f" % us,
          "TB/s %.2f" % (32 * D * C / us / 1e6), "frac %.3f" % (32 * D * C / us / 1e6 / 8.0))
