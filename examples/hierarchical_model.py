"""A USER'S hierarchical model on the library's fused kernels, two ways -- no build to write.

    theta = (mu, log tau, x_1 .. x_n);   x_d ~ N(mu, tau^2),  y_d ~ N(x_d, 1) observed;   mu ~ N(0, 25),  log tau ~ N(0, 1)

1. As a few lines of HIP C++ in the lane-spread form (`CTarget.from_source(form="lanes", head=2)`): the two hyper-parameters
   are the "head" every lane of a chain holds, the rows x_d are spread over the lanes of a wavefront, sums run in the library's
   fixed order.  hipcc compiles it when the object is built; the samplers then run a whole HMC trajectory / a whole
   delayed-rejection proposal as ONE launch with this density inlined (the kernels `bk.Funnel` itself uses).
2. A separable density written in PyTorch (`TorchModel(fn, D, compile=True)`): traced with torch.fx, differentiated symbolically
   and compiled the same way.

    python examples/hierarchical_model.py          # one MI355X
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "bayes-kit_amd")]

import torch

import bayes_kit_amd as bk

dev = torch.device("cuda", 0)
n, chains, draws = 64, 16384, 300
D = 2 + n
y = torch.zeros(D, dtype=torch.float64, device=dev)      # params[d] = y_d (the two head slots are unused)
y[2:] = 1.5 + 2.0 * torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))

HIER = """
template <class L>
__device__ double bk_lanes_density(L& c, const double* y) {
  const double mu = c.head(0), lt = c.head(1);
  const double it2 = exp(-2.0 * lt);                                        // 1 / tau^2
  const double n = (double)(c.dims() - 2);
  const double sq = c.sum([mu](double x, i64) { const double r = x - mu; return r * r; });
  const double sr = c.sum([mu](double x, i64) { return x - mu; });
  const double sy = c.sum([y](double x, i64 d) { const double r = y[d] - x; return r * r; });
  c.grad_head(0, it2 * sr - mu / 25.0);
  c.grad_head(1, (it2 * sq - n) - lt);
  c.grad([mu, it2, y](double x, i64 d) { return (y[d] - x) - it2 * (x - mu); });   // once, last
  return (((-0.5 * it2) * sq - n * lt) - 0.5 * sy) - (mu * mu / 50.0 + 0.5 * (lt * lt));
}
"""
model = bk.CTarget.from_source(HIER, D, params=y, form="lanes", head=2)

# Starts near the data (x = y, mu = mean y, tau = 1, jittered): N(0, I) starts -- the reference's default -- put a fraction of
# the chains deep in the neck of this centred parameterisation (tau = e^-3 with rows of unit size), where no step size of
# the ladder is stable and a chain stays put for the whole run.
g0 = torch.Generator().manual_seed(3)
init = torch.zeros((chains, D), dtype=torch.float64)
init[:, 0] = float(y[2:].mean()) + 0.1 * torch.randn(chains, generator=g0, dtype=torch.float64)
init[:, 1] = 0.1 * torch.randn(chains, generator=g0, dtype=torch.float64)
init[:, 2:] = y[2:].cpu() + 0.3 * torch.randn((chains, n), generator=g0, dtype=torch.float64)

# delayed-rejection HMC: every proposal (and its first ghost) one launch, the draw one hipGraph, no host synchronisation
dr = bk.DrGhmcDiag(model, 3, [0.3, 0.1, 0.03], [8, 16, 32], 0.2, chains=chains, seed=7, init=init)
print("DRGHMC: one launch per proposal:", dr._one_launch, "| host syncs per draw:", dr.host_syncs_per_draw)
for _ in range(draws):                                   # burn-in
    dr.advance()
mom = bk.RunningMoments(D, chains)
for _ in range(draws):
    theta, logp = dr.sample()
    mom.update(theta)
rh = torch.as_tensor(mom.rhat())
print(f"  R-hat: mu {float(rh[0]):.3f}  log tau {float(rh[1]):.3f}  max over rows {float(rh[2:].max()):.3f}")
print(f"  posterior mean of mu {float(theta[:, 0].mean()):.3f} (data mean {float(y[2:].mean()):.3f}), "
      f"of tau {float(theta[:, 1].exp().mean()):.3f} (data sd {float(y[2:].std()):.3f})")

# plain HMC on the same model: the whole trajectory one launch
hmc = bk.HMCDiag(model, 0.1, 16, chains=chains, seed=8, init=init)
for _ in range(50):
    hmc.sample()
print("HMC: whole trajectory in one launch:", hmc._lanes_traj, "| accept rate", round(hmc.accept_rate(), 3))

# a separable density from PyTorch code: traced, differentiated and compiled (no autograd at run time)
scale = torch.linspace(0.5, 3.0, 256, dtype=torch.float64, device=dev)
student = bk.TorchModel(lambda Th: (-2.5 * torch.log1p((Th / scale) ** 2 / 4.0)).sum(dim=1), 256, compile=True)
print("TorchModel(compile=True): compiled =", student.compiled is not None, "| note:", student.compile_note)
s = bk.HMCDiag(student, 0.5, 8, chains=chains, seed=9)
for _ in range(50):
    th, _ = s.sample()
print("  whole-draw kernel:", s._fused_draw, "| accept rate", round(s.accept_rate(), 3),
      "| sample sd of coordinate 0 / scale:", round(float(th[:, 0].std() / scale[0]), 2), "(t_4: 1.41)")
