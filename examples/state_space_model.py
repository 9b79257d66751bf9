"""A state-space model written in PyTorch with shifted slices, on the library's compiled per-chain kernels -- no HIP to write.

    theta = (a, log s, x_1 .. x_T);   phi = tanh(a);   x_1 ~ N(0, s^2 / (1 - phi^2)),  x_t ~ N(phi x_{t-1}, s^2);
    y_t ~ N(x_t, 0.5^2) observed;   a ~ N(0, 1),  log s ~ N(-1, 0.5^2)

`TorchModel(fn, D, compile=True)` reads the function once with torch.fx.  Its coordinates are coupled through the slices
`x[:, 1:]` and `x[:, :-1]`, so it is neither separable nor head-plus-exchangeable-rows: `trace_chain.py` differentiates it
symbolically (one derivative per slice offset) and emits the per-chain form, `CTarget.from_source` compiles it with hipcc, and the
samplers run the whole trajectory of every proposal as ONE launch (theta in a lane's registers, the momentum in LDS) -- lane
counts on the device, the draw one hipGraph, no host synchronisation, no autograd at run time.

    python examples/state_space_model.py          # one MI355X
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "bayes-kit_amd")]

import torch

import bayes_kit_amd as bk

dev = torch.device("cuda", 0)
T, chains, draws = 99, 8192, 400
D = T + 2
phi_true, s_true, obs_sd = 0.8, 0.5, 0.5
g = torch.Generator().manual_seed(11)
x_true = torch.zeros(T, dtype=torch.float64)
x_true[0] = s_true / (1 - phi_true ** 2) ** 0.5 * torch.randn((), generator=g, dtype=torch.float64)
for t in range(1, T):
    x_true[t] = phi_true * x_true[t - 1] + s_true * torch.randn((), generator=g, dtype=torch.float64)
y = (x_true + obs_sd * torch.randn(T, generator=g, dtype=torch.float64)).to(dev)


def log_density(Th):
    a, ls, x = Th[:, 0], Th[:, 1], Th[:, 2:]
    phi = torch.tanh(a)
    inn = x[:, 1:] - phi[:, None] * x[:, :-1]                              # the AR(1) innovations: two slices of different offsets
    prec = torch.exp(-2.0 * ls)
    ch = torch.cosh(a)                                                     # 1 - phi^2 = 1 / cosh(a)^2, without cancellation
    lp_x = -0.5 * prec * (inn * inn).sum(-1) - (T - 1) * ls \
        - 0.5 * prec / (ch * ch) * x[:, 0] ** 2 - ls - torch.log(ch)
    lp_y = -0.5 * (((y - x) / obs_sd) ** 2).sum(-1)
    return lp_x + lp_y - 0.5 * a * a - 0.5 * ((ls + 1.0) / 0.5) ** 2


t0 = time.perf_counter()
model = bk.TorchModel(log_density, D, compile=True)
print(f"TorchModel(compile=True): compiled form = {model.compiled_form} ({time.perf_counter() - t0:.1f} s incl. hipcc or cache)"
      f" | note: {model.compile_note}")

g0 = torch.Generator().manual_seed(3)
init = torch.zeros((chains, D), dtype=torch.float64)
init[:, 0] = 0.5 + 0.2 * torch.randn(chains, generator=g0, dtype=torch.float64)
init[:, 1] = -1.0 + 0.2 * torch.randn(chains, generator=g0, dtype=torch.float64)
init[:, 2:] = y.cpu() + 0.3 * torch.randn((chains, T), generator=g0, dtype=torch.float64)

dr = bk.DrGhmcDiag(model, 3, [0.05, 0.02, 0.008], [8, 16, 32], 0.2, chains=chains, seed=7, init=init)
print("DRGHMC: one launch per trajectory:", dr._traj_hook, "| host syncs per draw:", dr.host_syncs_per_draw)
for _ in range(draws):                                   # burn-in
    dr.advance()
torch.cuda.synchronize()
mom = bk.RunningMoments(D, chains)
t0 = time.perf_counter()
for _ in range(draws):
    theta, logp = dr.sample()
    mom.update(theta)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / draws
rh = torch.as_tensor(mom.rhat())
phi = torch.tanh(theta[:, 0])
print(f"  {ms:.2f} ms per draw of {chains} chains | chains with a non-finite state: {int((~torch.isfinite(theta).all(1)).sum())}")
print(f"  R-hat: a {float(rh[0]):.3f}  log s {float(rh[1]):.3f}  max over states {float(rh[2:].max()):.3f}")
print(f"  posterior mean of phi {float(phi.mean()):.3f} (sd {float(phi.std()):.3f}; truth {phi_true}), "
      f"of s {float(theta[:, 1].exp().mean()):.3f} (truth {s_true})")
rmse = float(((theta[:, 2:].mean(0).cpu() - x_true) ** 2).mean().sqrt())
print(f"  rmse of the posterior mean path against the true states {rmse:.3f} (observation noise {obs_sd})")

# the same function through autograd, for the price
auto = bk.DrGhmcDiag(bk.TorchModel(log_density, D), 3, [0.05, 0.02, 0.008], [8, 16, 32], 0.2, chains=chains, seed=7, init=init)
auto.sample()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    auto.sample()
torch.cuda.synchronize()
print(f"  the same function through autograd: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per draw, "
      f"{auto.host_syncs_per_draw} host syncs per draw")
