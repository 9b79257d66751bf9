"""Single-chain NumPy target densities used as oracle-side models (TEST INFRASTRUCTURE).

They satisfy the reference's structural Model protocol (``bayes_kit/typing.py:15-27``):
``dims()``, ``log_density(theta)``, ``log_density_gradient(theta)``.  The operation order
written here is the specification the built-in HIP targets (``bk_target_*`` in
``include/bkhip.h``) are tested against: every ``*`` and ``+`` individually rounded,
elementwise parts bit-exact, reductions within the stated tolerance.
"""
from __future__ import annotations

import numpy as np


class StdNormal:
    """D=1 standard normal; same arithmetic as the reference's test fixture
    ``test/models/std_normal.py:4-13``."""

    def dims(self) -> int:
        return 1

    def log_density(self, theta):
        t = theta[0]
        return -0.5 * t * t

    def log_density_gradient(self, theta):
        return -0.5 * theta[0] * theta[0], -theta


class IsoGaussian:
    """logp = -1/2 theta.theta ; grad = -theta   (BASELINE.json config 2)."""

    def __init__(self, D: int):
        self._D = int(D)

    def dims(self) -> int:
        return self._D

    def log_density(self, theta):
        return -0.5 * np.dot(theta, theta)

    def log_density_gradient(self, theta):
        return -0.5 * np.dot(theta, theta), -theta


class DiagGaussian:
    """logp = -1/2 sum_i lam_i theta_i^2 ; grad = -(lam * theta)   (config 3)."""

    def __init__(self, lam):
        self._lam = np.asarray(lam, dtype=np.float64)

    def dims(self) -> int:
        return self._lam.shape[0]

    def log_density(self, theta):
        t = self._lam * theta
        return -0.5 * np.dot(theta, t)

    def log_density_gradient(self, theta):
        t = self._lam * theta
        return -0.5 * np.dot(theta, t), -t


class Funnel:
    """Neal's funnel, v = theta[0] ~ N(0, 3^2), x_i ~ N(0, e^v), i = 1..D-1 (config 4).

    logp = -v^2/18 - (n/2) v - 1/2 e^{-v} sum x^2,  n = D-1
    d/dv = -v/9 - n/2 + 1/2 e^{-v} sum x^2 ;  d/dx_i = -e^{-v} x_i
    """

    def __init__(self, D: int):
        self._D = int(D)

    def dims(self) -> int:
        return self._D

    def _parts(self, theta):
        v = theta[0]
        x = theta[1:]
        ev = np.exp(-v)
        s = np.dot(x, x)
        hn = 0.5 * (self._D - 1)
        he = 0.5 * ev
        return v, x, ev, s, hn, he

    def log_density(self, theta):
        v, x, ev, s, hn, he = self._parts(theta)
        return ((-(v * v) / 18.0) - hn * v) - he * s

    def log_density_gradient(self, theta):
        v, x, ev, s, hn, he = self._parts(theta)
        lp = ((-(v * v) / 18.0) - hn * v) - he * s
        g = np.empty(self._D, dtype=np.float64)
        g[0] = ((-v / 9.0) - hn) + he * s
        g[1:] = -(ev * x)
        return lp, g


class LogisticRegression:
    """y_n ~ Bernoulli(sigmoid(x_n . theta)), theta ~ N(0, s^2 I) (BASELINE.json config 5).
    No reference counterpart: specification of the product's MFMA target (tolerance parity)."""

    def __init__(self, X, y, prior_scale=1.0):
        self._X = np.asarray(X, dtype=np.float64)
        self._y = np.asarray(y, dtype=np.float64)
        self._inv_s2 = 1.0 / prior_scale**2

    def dims(self) -> int:
        return self._X.shape[1]

    def log_likelihood(self, theta):
        z = self._X @ theta
        return np.sum(self._y * z - np.logaddexp(0.0, z))

    def log_prior(self, theta):
        return -0.5 * self._inv_s2 * np.dot(theta, theta)

    def log_density(self, theta):
        return self.log_likelihood(theta) + self.log_prior(theta)

    def log_density_gradient(self, theta):
        z = self._X @ theta
        p = 1.0 / (1.0 + np.exp(-z))
        g = self._X.T @ (self._y - p) - self._inv_s2 * theta
        return np.sum(self._y * z - np.logaddexp(0.0, z)) + self.log_prior(theta), g


class GaussPriorLik:
    """A ``LogPriorLikelihoodModel`` (``bayes_kit/typing.py:37-42``) with elementwise parts, for the SMC fixtures:
    prior theta ~ N(0, s0^2 I), likelihood y_d ~ N(theta_d, 1 / prec_d).

    log_prior = (-0.5 / s0^2) * sum theta_d^2 ;  log_likelihood = -0.5 * sum prec_d (theta_d - y_d)^2
    (sums are ``np.sum`` of the elementwise products)."""

    def __init__(self, y, prec, prior_scale=1.0):
        self._y = np.asarray(y, dtype=np.float64)
        self._prec = np.asarray(prec, dtype=np.float64)
        self._c0 = -0.5 / prior_scale**2

    def dims(self) -> int:
        return self._y.shape[0]

    def log_prior(self, theta):
        return self._c0 * np.sum(theta * theta)

    def log_likelihood(self, theta):
        r = theta - self._y
        return -0.5 * np.sum(self._prec * (r * r))

    def log_density(self, theta):
        return self.log_likelihood(theta) + self.log_prior(theta)

    def posterior_mean(self):
        p0 = -2.0 * self._c0
        return self._prec * self._y / (self._prec + p0)

    def posterior_var(self):
        return 1.0 / (self._prec - 2.0 * self._c0)


class FunnelCanonical(Funnel):
    """The funnel with sum x^2 taken in the HIP library's canonical order (csrc/bk_lanes.hpp: 16 interleaved class sums
    -> 4 group sums -> total) instead of ``np.dot``: class c = (d - 1) mod 16 of row d >= 1 is summed in row order,
    q[g] = ((cs[g] + cs[g+4]) + cs[g+8]) + cs[g+12], s = ((q[0] + q[1]) + q[2]) + q[3].  With this model the oracle and
    the device evaluate the SAME sequence of rounded operations except inside exp(); the comparison between them is
    then held to SURVEY 8c's 1e-9 over every draw, and the chaos-widened bound stays where it belongs: between this
    order and the reference's own ``np.dot`` (tests/test_oracle_golden.py).

    exp() is the library's ``bk_exp`` (``oracle.rng.exp_bk``): with the summation order AND the exponential shared,
    the device and this oracle run the same sequence of rounded operations and agree BIT FOR BIT."""

    def _parts(self, theta):
        v = theta[0]
        x = theta[1:]
        cs = [0.0] * 16
        for i, xi in enumerate(x):
            cs[i % 16] = cs[i % 16] + xi * xi
        q = [((cs[g] + cs[g + 4]) + cs[g + 8]) + cs[g + 12] for g in range(4)]
        s = ((q[0] + q[1]) + q[2]) + q[3]
        from .rng import exp_bk

        ev = exp_bk(-v)
        hn = 0.5 * (self._D - 1)
        he = 0.5 * ev
        return v, x, ev, s, hn, he
