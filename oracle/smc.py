"""NumPy restatement of the reference's likelihood-tempered SMC (TEST INFRASTRUCTURE: only tests/, smoke() and the
cpu_baseline leg of bench.py may import this package).

Follows ``bayes_kit/smc.py``: ``TemperedLikelihoodSMC.__init__`` (:13-27), ``time`` (:42-43), ``transition``
(:45-61: every particle through the move kernel at the PREVIOUS temperature, in particle order, then the
reweighting), ``importance_resample`` (:64-75: ``exp(lp - lpminus1)``, ``choice(p = w / w.sum())``, ``thetas[idxs]``)
and ``metropolis_kernel`` (:79-89: ``normal(loc=theta, scale)`` then ``log(uniform()) < lp(theta*) - lp(theta)``).

The reference draws from the process-global ``numpy.random`` -- seedable with ``np.random.seed(s)`` -- and consumes,
per temperature: for particle 0..M-1 {D normals, 1 uniform}, then the M uniforms of ``choice``.  Here the stream is an
explicit object with ``normal(loc, scale)``, ``uniform()`` and ``choice_uniforms(m)``: ``oracle.rng.LegacyStream``
(the restated MT19937 + legacy distributions), a wrapper around a real ``numpy.random.RandomState``, or a replay of
recorded values.

**Parity status: pinned** -- ``tests/golden/smc_*.npz`` hold the particles after every move and every resampling, the
ancestor indices and the consumed stream values of the real reference run under ``np.random.seed``
(``tests/golden/make_golden.py::run_smc_case``); ``tests/test_oracle_golden.py`` checks this file against them bit
for bit.
"""
from __future__ import annotations

import numpy as np

from .rng import legacy_choice, numpy_pairwise_sum


class NumpyLegacySource:
    """The same stream interface over a real ``numpy.random.RandomState`` (or the ``numpy.random`` module itself)."""

    def __init__(self, rs):
        self._rs = rs

    def normal(self, loc, scale):
        return self._rs.normal(loc=loc, scale=scale)

    def uniform(self):
        return self._rs.uniform()

    def choice_uniforms(self, m):
        return self._rs.random_sample(m)


class ReplaySource:
    """Replays recorded standard normals / uniforms in consumption order."""

    def __init__(self, normals, uniforms, choice_uniforms):
        self._z = np.asarray(normals, dtype=np.float64).reshape(-1)
        self._u = np.asarray(uniforms, dtype=np.float64).reshape(-1)
        self._c = np.asarray(choice_uniforms, dtype=np.float64).reshape(-1)
        self._iz = self._iu = self._ic = 0

    def normal(self, loc, scale):
        loc = np.atleast_1d(np.asarray(loc, dtype=np.float64))
        z = self._z[self._iz:self._iz + loc.shape[0]]
        self._iz += loc.shape[0]
        return loc + scale * z

    def uniform(self):
        self._iu += 1
        return float(self._u[self._iu - 1])

    def choice_uniforms(self, m):
        self._ic += m
        return self._c[self._ic - m:self._ic]


def metropolis_kernel(scale, stream):
    """smc.py:79-89 with the stream explicit."""

    def move(theta, lp):
        theta_star = stream.normal(theta, scale)
        if np.log(stream.uniform()) < lp(theta_star) - lp(theta):
            return theta_star
        return theta

    return move


def importance_resample(thetas, lpminus1, lp, stream):
    """smc.py:64-75.  Returns (resampled particles, ancestor indices, weights)."""
    a = np.array([lp(th) for th in thetas])
    b = np.array([lpminus1(th) for th in thetas])
    weights = np.exp(a - b)
    p = weights / numpy_pairwise_sum(weights)  # np.sum
    idxs = legacy_choice(p, stream.choice_uniforms(thetas.shape[0]))
    return thetas[idxs], idxs, weights


class TemperedLikelihoodSMC:
    def __init__(self, model, M, N, sample_initial, kernel, stream):
        self.M, self.N = M, N
        self.thetas = np.array([np.asarray(sample_initial(i), dtype=np.float64) for i in range(M)])
        self.D = self.thetas.shape[1]
        self._model, self.kernel, self._stream = model, kernel, stream
        self.moved, self.idxs, self.weights = None, None, None  # of the last transition

    def time(self, n):
        return n / self.N

    def _tempered(self, t):
        m = self._model
        return lambda theta: m.log_likelihood(theta) * t + m.log_prior(theta)

    def transition(self, n):
        lpminus1, lp = self._tempered(self.time(n - 1)), self._tempered(self.time(n))
        for m in range(self.M):
            self.thetas[m] = self.kernel(np.atleast_1d(self.thetas[m]), lpminus1)
        self.moved = self.thetas.copy()
        self.thetas, self.idxs, self.weights = importance_resample(self.thetas, lpminus1, lp, self._stream)

    def run(self):
        for n in range(1, self.N + 1):
            self.transition(n)
