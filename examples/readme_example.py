"""The README example of flatironinstitute/bayes-kit (README.md:13-32) on bayes_kit_amd, then
the same sampler with 65,536 chains.  Run on an MI355X:  python examples/readme_example.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "bayes-kit_amd")]

import numpy as np

import bayes_kit_amd as bk


class StdNormal:  # a reference-style model: NumPy in, NumPy out, one chain
    def dims(self):
        return 1

    def log_density(self, theta):
        return -0.5 * theta[0] * theta[0]

    def log_density_gradient(self, theta):
        return -0.5 * theta[0] * theta[0], -theta


# 1) drop-in, exactly as with the reference (integrator, RNG and accept run on the GPU; with an
#    int seed the stream is the reference's own PCG64, so the draws are the reference's draws)
import time

sampler = bk.MALA(StdNormal(), 0.2, seed=12345)
sampler.sample()  # (first launch: code objects)
t0 = time.perf_counter()
draws = np.array([sampler.sample()[0] for _ in range(1000)])
us = 1e6 * (time.perf_counter() - t0) / 1000
print(f"1 chain   : mean {draws.mean():+.3f}  var {draws.var(ddof=1):.3f}   {us:.1f} us per draw ({sampler.path})")

# 2) many chains: a device-resident batched model, one chain per GPU lane
sampler = bk.MALA(bk.IsoGaussian(1), 0.2, chains=65536, seed=12345)
for _ in range(200):
    theta, logp = sampler.sample()  # (65536, 1) and (65536,) device tensors
print(f"65536 chains after 200 draws: mean {theta.mean().item():+.4f}  var {theta.var().item():.4f}  "
      f"accept {sampler.accept_rate():.2f}")

# 3) HMC on an ill-conditioned Gaussian, R-hat over all chains from streaming moments
lam = np.logspace(0, 1, 64)  # (static HMC: keep sqrt(lam)*eps*L away from multiples of pi)
hmc = bk.HMCDiag(bk.DiagGaussian(lam), 0.08, 10, chains=4096, seed=1)
mom = bk.RunningMoments(64, 4096)
for _ in range(100):
    theta, _ = hmc.sample()
    mom.update(theta)
print(f"HMC: accept {hmc.accept_rate():.2f}  max R-hat {mom.rhat().max():.4f}")
