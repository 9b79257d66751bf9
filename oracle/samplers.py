"""Single-chain NumPy restatement of the reference samplers (TEST INFRASTRUCTURE).

Each class follows the arithmetic and the RNG-consumption order of the reference file it
cites, line by line in behaviour but written in this repo's own structure (explicit
phase-space records instead of the reference's gradient-cache stack; see SURVEY.md
section 3.3.1).  ``tests/test_oracle_golden.py`` pins them bit-for-bit (theta, returned
logp and the bit generator's final state) against vectors produced by the imported
reference (``tests/golden/make_golden.py``).

One deliberate widening: ``metric_diag`` may be a length-D array here.  The reference
evaluates ``metric_diag or np.ones(D)`` (``bayes_kit/hmc.py:22``, ``drghmc.py:70``), which
raises for any D>1 array, so it can only be given a metric by assigning ``_metric`` after
construction; the golden generator does exactly that.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence

import numpy as np


def _init_theta(init, rng, D):
    # bayes_kit/hmc.py:24-28, mala.py:26-30, drghmc.py:72-76 -- an empty init is "absent"
    if init is not None and init.shape != (0,):
        return init
    return rng.normal(size=D)


class HMCDiag:
    """bayes_kit/hmc.py:8-63."""

    def __init__(self, model, stepsize, steps, metric_diag=None, init=None, seed=None):
        self._model = model
        self._dim = model.dims()
        self._stepsize = stepsize
        self._steps = steps
        self._metric = (
            np.ones(self._dim) if metric_diag is None else np.asarray(metric_diag, dtype=np.float64)
        )
        self._rng = np.random.default_rng(seed)  # hmc.py:23
        self._theta = _init_theta(init, self._rng, self._dim)
        self.last_accept = False

    def __iter__(self):
        return self

    def __next__(self):
        return self.sample()

    def _joint(self, theta, rho):
        # hmc.py:36-38
        kin = 0.5 * np.dot(rho, self._metric * rho)
        return self._model.log_density(theta) - kin

    def sample(self):
        eps, m = self._stepsize, self._metric
        rho = self._rng.normal(size=self._dim)  # hmc.py:56
        h0 = self._joint(self._theta, rho)  # hmc.py:57
        # hmc.py:40-53: back half step, L x (kick, drift, gradient), forward half step
        theta = self._theta
        _, g = self._model.log_density_gradient(theta)
        half = 0.5 * eps
        r = rho - half * np.multiply(m, g)
        for _ in range(self._steps):
            r = r + eps * np.multiply(m, g)
            theta = theta + eps * r
            _, g = self._model.log_density_gradient(theta)
        rho1 = r + half * np.multiply(m, g)
        h1 = self._joint(theta, rho1)  # hmc.py:59
        self.last_accept = bool(np.log(self._rng.uniform()) < h1 - h0)  # hmc.py:60
        if self.last_accept:
            self._theta = theta
            return self._theta, h1
        return self._theta, h0


class HMCDense(HMCDiag):
    """HMC with a dense metric: NO reference counterpart (parity unpinned).  Specification of
    the product's MFMA path, in the reference's (theta, velocity) variables (hmc.py:40-63 shape):
    M = velocity covariance;  rho = chol(M) @ z, z = rng.normal(size=D);  kick rho += eps*(M @ grad);
    drift theta += eps*rho;  kinetic energy 1/2 rho.(M^-1 @ rho).  Equals HMCDiag for M = I."""

    def __init__(self, model, stepsize, steps, metric_dense, init=None, seed=None):
        super().__init__(model, stepsize, steps, metric_diag=None, init=init, seed=seed)
        M = np.asarray(metric_dense, dtype=np.float64)
        self._M = 0.5 * (M + M.T)
        self._L = np.linalg.cholesky(self._M)
        Mi = np.linalg.inv(self._M)
        self._Minv = 0.5 * (Mi + Mi.T)

    def _joint(self, theta, rho):
        return self._model.log_density(theta) - 0.5 * np.dot(rho, self._Minv @ rho)

    def sample(self):
        eps, M = self._stepsize, self._M
        rho = self._L @ self._rng.normal(size=self._dim)
        h0 = self._joint(self._theta, rho)
        theta = self._theta
        _, g = self._model.log_density_gradient(theta)
        half = 0.5 * eps
        r = rho - half * (M @ g)
        for _ in range(self._steps):
            r = r + eps * (M @ g)
            theta = theta + eps * r
            _, g = self._model.log_density_gradient(theta)
        rho1 = r + half * (M @ g)
        h1 = self._joint(theta, rho1)
        self.last_accept = bool(np.log(self._rng.uniform()) < h1 - h0)
        if self.last_accept:
            self._theta = theta
            return self._theta, h1
        return self._theta, h0


class MALA:
    """bayes_kit/mala.py:14-79 with the accept rule of metropolis.py:41-76."""

    def __init__(self, model, epsilon, init=None, seed=None):
        self._model = model
        self._epsilon = epsilon
        self._dim = model.dims()
        self._rng = np.random.default_rng(seed)
        self._theta = _init_theta(init, self._rng, self._dim)
        lp, g = model.log_density_gradient(self._theta)  # mala.py:31-32
        self._lp, self._grad = lp, np.asanyarray(g)
        self.last_accept = False

    def __iter__(self):
        return self

    def __next__(self):
        return self.sample()

    def _logq(self, to, frm, grad_frm):
        # mala.py:68-79
        x = to - frm - self._epsilon * grad_frm
        return (-0.25 / self._epsilon) * x.dot(x)

    def sample(self):
        eps = self._epsilon
        z = self._rng.normal(size=self._model.dims())
        prop = self._theta + eps * self._grad + np.sqrt(2 * eps) * z  # mala.py:41-45
        lp_prop, g_prop = self._model.log_density_gradient(prop)
        g_prop = np.asanyarray(g_prop)
        fwd = self._logq(prop, self._theta, self._grad)
        rev = self._logq(self._theta, prop, g_prop)
        # metropolis.py:70-76
        ratio = (lp_prop - self._lp) + (rev - fwd)
        self.last_accept = bool(np.log(self._rng.uniform()) < ratio)
        if self.last_accept:
            self._theta, self._lp, self._grad = prop, lp_prop, g_prop
        return self._theta, self._lp


def metropolis_accept_test(lp_proposal, lp_current, rng):
    """bayes_kit/metropolis.py:12-38."""
    return bool(np.log(rng.uniform()) < lp_proposal - lp_current)


def metropolis_hastings_accept_test(lp_proposal, lp_current, lp_forward, lp_reverse, rng):
    """bayes_kit/metropolis.py:41-76."""
    ratio = (lp_proposal - lp_current) + (lp_reverse - lp_forward)
    return bool(np.log(rng.uniform()) < ratio)


class MetropolisHastings:
    """bayes_kit/metropolis.py:79-135: user proposal + user transition log density."""

    def __init__(self, model, proposal_fn, transition_lp_fn, *, init=None, seed=None):
        self._model = model
        self._dim = model.dims()
        self._rng = np.random.default_rng(seed)  # metropolis.py:91
        self._proposal_fn = proposal_fn
        self._transition_lp_fn = transition_lp_fn
        self._theta = _init_theta(init, self._rng, self._dim)
        self._log_p_theta = model.log_density(self._theta)  # metropolis.py:99
        self.last_accept = False

    def __iter__(self):
        return self

    def __next__(self):
        return self.sample()

    def _accepts(self, prop, lp_prop):
        fwd = self._transition_lp_fn(prop, self._theta)  # metropolis.py:126
        rev = self._transition_lp_fn(self._theta, prop)  # metropolis.py:127
        return metropolis_hastings_accept_test(lp_prop, self._log_p_theta, fwd, rev, self._rng)

    def sample(self):
        prop = np.asanyarray(self._proposal_fn(self._theta), dtype=np.float64)  # metropolis.py:114-116
        lp_prop = self._model.log_density(prop)
        self.last_accept = self._accepts(prop, lp_prop)
        if self.last_accept:
            self._theta, self._log_p_theta = prop, lp_prop
        return self._theta, self._log_p_theta


class Metropolis(MetropolisHastings):
    """bayes_kit/metropolis.py:138-155: symmetric proposal, plain Metropolis ratio."""

    def __init__(self, model, proposal_fn, *, init=None, seed=None):
        super().__init__(model, proposal_fn, None, init=init, seed=seed)

    def _accepts(self, prop, lp_prop):
        return metropolis_accept_test(lp_prop, self._log_p_theta, self._rng)


class _Point(NamedTuple):
    """A phase-space point with the model outputs that belong to it."""

    theta: np.ndarray
    rho: np.ndarray
    logp: float
    grad: np.ndarray


class DrGhmcDiag:
    """bayes_kit/drghmc.py:10-446 in stack-free form.

    ``last_schedule`` records, per draw, the tags of the leapfrog trajectories that were
    run ("P<k>" proposals, "G<i>" ghost proposals, nesting shown by parentheses) and
    ``last_grad_evals`` the number of gradient evaluations - an exact control-flow
    fixture next to theta.
    """

    def __init__(
        self,
        model,
        max_proposals,
        leapfrog_step_sizes: Sequence[float],
        leapfrog_step_counts: Sequence[int],
        damping,
        metric_diag=None,
        init=None,
        seed=None,
        prob_retry: bool = True,
    ):
        self._model = model
        self._dim = model.dims()
        self._max_proposals = max_proposals
        self._leapfrog_step_sizes = leapfrog_step_sizes
        self._leapfrog_step_counts = leapfrog_step_counts
        self._damping = damping
        self._metric = (
            np.ones(self._dim) if metric_diag is None else np.asarray(metric_diag, dtype=np.float64)
        )
        self._rng = np.random.default_rng(seed)
        self._theta = _init_theta(init, self._rng, self._dim)
        self._rho = self._rng.normal(size=self._dim)  # drghmc.py:77 (after theta)
        self._prob_retry = prob_retry
        self._cur_model: Optional[tuple] = None  # (logp, grad) at _theta, once known
        self.last_schedule: list = []
        self.last_grad_evals = 0
        self.last_accept_stage = -1

    def __iter__(self):
        return self

    def __next__(self):
        return self.sample()

    # -- pieces -----------------------------------------------------------------
    def _grad(self, theta):
        self.last_grad_evals += 1
        lp, g = self._model.log_density_gradient(theta)
        return lp, np.asanyarray(g)

    def _joint(self, pt: _Point):
        # drghmc.py:249-251
        potential = -pt.logp
        kinetic = 0.5 * np.dot(pt.rho, self._metric * pt.rho)
        return -(potential + kinetic)

    def _retry(self, reject_logp):
        return self._prob_retry * reject_logp  # drghmc.py:317

    def _proposal_map(self, pt: _Point, k: int, tag: str) -> _Point:
        # drghmc.py:253-289 then the flip of :345
        self.last_schedule.append(tag)
        h, n = self._leapfrog_step_sizes[k], self._leapfrog_step_counts[k]
        m = self._metric
        theta = np.array(pt.theta, copy=True)
        r = pt.rho + 0.5 * h * np.multiply(m, pt.grad).squeeze()
        theta += h * r
        for _ in range(n - 1):
            _, g = self._grad(theta)
            r += h * np.multiply(m, g).squeeze()
            theta += h * r
        lp, g = self._grad(theta)
        rho = r + 0.5 * h * np.multiply(m, g).squeeze()
        return _Point(theta, -rho, lp, g)

    def _accept(self, prop: _Point, k: int, cur_hastings, cur_logp, tag: str):
        # drghmc.py:391-446
        prop_logp = self._joint(prop)
        prop_hastings = 0
        for i in range(k):
            ghost = self._proposal_map(prop, i, "G%d(%s)" % (i, tag))
            a, _ = self._accept(ghost, i, prop_hastings, prop_logp, "G%d(%s)" % (i, tag))
            if a == 0:
                return -np.inf, prop_logp
            prop_hastings += np.log1p(-np.exp(a))
        frac = (
            (prop_logp - cur_logp)
            + (prop_hastings - cur_hastings)
            + (self._retry(prop_hastings) - self._retry(cur_hastings))
        )
        return min(0, frac), prop_logp

    # -- one draw ---------------------------------------------------------------
    def sample(self):
        self.last_schedule = []
        self.last_grad_evals = 0
        self.last_accept_stage = -1
        self._rho = self._rng.normal(  # drghmc.py:360-364
            loc=self._rho * np.sqrt(1 - self._damping),
            scale=np.sqrt(self._damping),
            size=self._dim,
        )
        if self._cur_model is None:  # drghmc.py:243-245, first draw only
            self._cur_model = self._grad(self._theta)
        cur = _Point(self._theta, self._rho, *self._cur_model)
        cur_logp = self._joint(cur)
        cur_hastings, reject_logp = 0.0, 0.0
        for k in range(self._max_proposals):
            if not np.log(self._rng.uniform()) < self._retry(reject_logp):  # :369-371
                break
            prop = self._proposal_map(cur, k, "P%d" % k)
            accept_logp, prop_logp = self._accept(prop, k, cur_hastings, cur_logp, "P%d" % k)
            if np.log(self._rng.uniform()) < accept_logp:  # :378-381
                cur, cur_logp = prop, prop_logp
                self.last_accept_stage = k
                break
            reject_logp = np.log1p(-np.exp(accept_logp))  # :383-384
            cur_hastings += reject_logp
        self._theta, self._cur_model = cur.theta, (cur.logp, cur.grad)
        self._rho = -cur.rho  # drghmc.py:388
        return self._theta, cur_logp
