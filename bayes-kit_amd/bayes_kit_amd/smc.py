"""Likelihood-tempered sequential Monte Carlo on the GPU: many-particle counterpart of
``bayes_kit/smc.py:12-89`` (``TemperedLikelihoodSMC``, ``importance_resample``,
``metropolis_kernel``).

Same constructor shape ``(model, M, N, sample_initial, kernel)``, ``run()``,
``transition(n)``, ``time(n)``, ``thetas`` and iteration protocol.  The M particles live in
one chain-contiguous ``[D, M]`` buffer (one particle per GPU lane); a temperature step is

1. the move kernel applied to every particle at temperature (n-1)/N       [smc.py:53-57]
2. weights exp(lp_n - lp_{n-1}), multinomial resampling                   [smc.py:60, 64-75]

Randomness, two modes:

* default -- every particle slot owns a Philox stream (key = (seed, slot)): proposal normals, the
  accept uniform and one resampling uniform per slot, generated on the device; reproducible, and
  the same at any number of ranks.
* **reference stream** (``seed=np.random``, a ``numpy.random.RandomState``, or any object with its
  ``standard_normal(size=) / uniform() / random_sample(n)``): the reference draws from the
  process-global legacy stream (smc.py:73, 81, 85), which ``np.random.seed(s)`` seeds.  In this
  mode the host takes the values from that stream in the reference's own order -- per temperature,
  for particle 0..M-1 {D normals, 1 uniform}, then ``choice``'s M uniforms -- and everything else
  (proposal, densities, accept test, weights, ``choice``'s cdf and search, the gather) runs on the
  device with the reference's arithmetic: after ``np.random.seed(s)`` the particles and ancestor
  indices equal the reference's bit for bit (``tests/golden/smc_*.npz``).  Single rank,
  ``metropolis_kernel`` (the reference's only move kernel).

Model: the batched form of ``LogPriorLikelihoodModel`` (typing.py:37-42):
``log_prior(Theta) -> (M,)``, ``log_likelihood(Theta) -> (M,)`` on a (M, D) device view; for
the gradient-based move kernel also ``log_density_gradient_tempered(Theta, t)``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from ._engine import make_streams


class _RWMKernel:
    """Random-walk Metropolis move, the reference's ``metropolis_kernel(scale)`` (smc.py:79-89)."""

    def __init__(self, scale: float):
        self.scale = float(scale)

    def move(self, smc, t: float) -> None:
        ops = smc._ops
        th, prop = smc._theta_dc, smc._prop_dc
        lp_cur = smc._tempered(th, t)
        if smc._ref_stream is not None:
            # the reference's stream in the reference's order: particle m takes D normals (smc.py:81), then one
            # uniform (smc.py:85), before particle m + 1 takes any
            rs, M, D = smc._ref_stream, smc.M, smc.D
            z, u = np.empty((M, D)), np.empty(M)
            for m in range(M):
                z[m] = rs.standard_normal(size=D)
                u[m] = rs.uniform()
            z_dc = torch.as_tensor(np.ascontiguousarray(z.T)).to(th.device)
            # normal(loc=theta, scale) = loc + scale * gauss: a rounded product, then a rounded sum
            torch.add(th, z_dc * self.scale, out=prop)
            smc._logu.copy_(torch.as_tensor(np.log(u)))
            lp_prop = smc._tempered(prop, t)
            ops.mh_accept(_lib.ACCEPT_MALA, lp_cur, None, lp_prop, None, smc._logu, smc._mask, None, smc._accepted)
            ops.select_columns(smc._mask, th, prop)
            return
        # theta* = normal(loc=theta, scale) [smc.py:81]
        ops.momentum_refresh(smc._rng_kind, smc._rng_state, th, 1.0, self.scale, prop, None, None,
                             None, getattr(smc, "_rng_work", None))
        lp_prop = smc._tempered(prop, t)
        ops.log_uniform(smc._rng_kind, smc._rng_state, smc._logu)
        # accept iff log u < lp(theta*) - lp(theta) [smc.py:85]
        ops.mh_accept(_lib.ACCEPT_MALA, lp_cur, None, lp_prop, None, smc._logu, smc._mask, None, smc._accepted)
        ops.select_columns(smc._mask, th, prop)


class _MALAKernel:
    """Langevin move (mala.py:40-66 arithmetic) on the tempered density, `steps` times."""

    def __init__(self, epsilon: float, steps: int = 1):
        self.epsilon, self.steps = float(epsilon), int(steps)

    def move(self, smc, t: float) -> None:
        ops, eps = smc._ops, self.epsilon
        th, prop = smc._theta_dc, smc._prop_dc
        f64 = dict(dtype=torch.float64, device=th.device)
        lp, g = smc._model.log_density_gradient_tempered(th.t(), t)
        grad = torch.empty_like(th)
        ops.relayout(g.t(), grad)
        lp = lp.clone()
        grad_p = torch.empty_like(th)
        fwd, rev = torch.empty(smc.M, **f64), torch.empty(smc.M, **f64)
        for _ in range(self.steps):
            ops.mala_propose(smc._rng_kind, smc._rng_state, th, grad, prop, eps, math.sqrt(2 * eps))
            lp_p, g_p = smc._model.log_density_gradient_tempered(prop.t(), t)
            ops.relayout(g_p.t(), grad_p)
            ops.mala_logq(th, grad, prop, grad_p, eps, fwd, rev)
            ops.log_uniform(smc._rng_kind, smc._rng_state, smc._logu)
            ops.mh_accept(_lib.ACCEPT_MALA, lp, fwd, lp_p.contiguous(), rev, smc._logu, smc._mask, None,
                          smc._accepted)
            ops.select_columns(smc._mask, th, prop, grad, grad_p)


class _TemperedTarget:
    """The tempered density log_likelihood * t + log_prior as a batched GradModel.  For models with the engine's
    fast path and a temperature argument (``bk_eval(theta, grad, logp, t, loglik_out)``: bk.LogisticRegression) the
    sampler's calls go straight to it -- a leapfrog step then forms no log density at all -- and every call that does
    form one also leaves the particles' untempered log likelihood in ``ll_last`` (what the next reweighting needs)."""

    batched = True

    def __init__(self, model):
        self._model, self.t = model, 1.0
        self.ll_last = None
        self.gll_last = None   # the likelihood's gradient at the points of the last evaluation that formed a log density
        self.track_gradient = hasattr(model, "bk_retemper")
        if hasattr(model, "bk_eval") and hasattr(model, "log_density_gradient_tempered") and hasattr(model, "log_likelihood"):
            self.bk_eval = self._bk_eval

    def dims(self):
        return self._model.dims()

    def log_density(self, Theta):
        return self._model.log_likelihood(Theta) * self.t + self._model.log_prior(Theta)

    def log_density_gradient(self, Theta):
        return self._model.log_density_gradient_tempered(Theta, self.t)

    def _bk_eval(self, theta_dc, grad_out, logp_out):
        ll = gll = None
        if logp_out is not None:
            ll = torch.empty(theta_dc.shape[1], dtype=torch.float64, device=theta_dc.device)
            if self.track_gradient and grad_out is not None:
                if self.gll_last is None or self.gll_last.shape != theta_dc.shape or self.gll_last.stride() != theta_dc.stride():
                    self.gll_last = torch.empty_strided(theta_dc.shape, theta_dc.stride(), dtype=torch.float64,
                                                        device=theta_dc.device)
                gll = self.gll_last
        if gll is not None:
            self._model.bk_eval(theta_dc, grad_out, logp_out, self.t, ll, gll)
        else:
            self._model.bk_eval(theta_dc, grad_out, logp_out, self.t, ll)
        if ll is not None:
            self.ll_last = ll


class _HMCKernel:
    """`draws` HMC transitions (bayes_kit/hmc.py:55-63 arithmetic, through HMCDiag) on the tempered
    density; optionally with a diagonal or a dense metric (dense = fp64 MFMA GEMMs).

    adapt_metric (extension): before the moves of every temperature the dense metric is set to the particles' own
    per-coordinate variance (over all ranks) -- the scale of the tempered posterior shrinks by orders of magnitude
    along the ladder (config 5: prior width 1 -> posterior width 0.05), and a fixed metric either barely moves the
    early particles or rejects every late proposal."""

    def __init__(self, stepsize, steps, draws=1, metric_diag=None, metric_dense=None, adapt_metric=False):
        self.stepsize, self.steps, self.draws = float(stepsize), int(steps), int(draws)
        self.metric_diag, self.metric_dense = metric_diag, metric_dense
        self.adapt_metric = bool(adapt_metric)
        self._hmc = None
        self._target = None
        self._ll = None
        self._ll_cur, self._gll, self._parts_ok = None, None, False   # untempered (loglik, its gradient) of the particles
        self.accept_rates = []

    def _particle_variance(self, smc):
        th = smc._theta_dc
        n = torch.tensor([float(th.shape[1])], dtype=torch.float64, device=th.device)
        stats = torch.cat([th.sum(dim=1), (th * th).sum(dim=1), n])
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(smc._group) > 1:
            from . import dist as _bkdist

            stats = sum(_bkdist.all_gather(stats, smc._group))
        D = th.shape[0]
        cnt = stats[-1]
        mean = stats[:D] / cnt
        var = (stats[D:2 * D] / cnt - mean * mean).clamp(min=1e-300) * (cnt / (cnt - 1.0).clamp(min=1.0))
        return var

    def move(self, smc, t: float) -> None:
        from .hmc import HMCDiag

        if self._hmc is None:
            self._target = _TemperedTarget(smc._model)
            base = int(smc._rng_state[0, 0].item()) & ((1 << 63) - 1)  # the SMC's Philox key word
            md = self.metric_dense
            if self.adapt_metric and md is None and self.metric_diag is None:
                md = torch.eye(smc.D, dtype=torch.float64)
            self._hmc = HMCDiag(self._target, self.stepsize, self.steps, metric_diag=self.metric_diag,
                                init=smc.thetas, seed=base ^ 0x5DEECE66D, chains=smc.M, chain_id0=smc._slot0,
                                metric_dense=md, graph=False, ops=smc._ops,
                                prefetch_rng=False if self.adapt_metric else None)
        h = self._hmc
        if self.adapt_metric and h._M is not None:
            h.set_metric_dense(self._particle_variance(smc))   # (1-D: a diagonal metric, installed on the device)
        self._target.t = float(t)
        h._theta_dc.copy_(smc._theta_dc)
        track = hasattr(self._target, "bk_eval")
        tg = track and self._target.track_gradient
        if tg and self._parts_ok:
            # new temperature, same points (resampled: the tracked parts were gathered with them): the tempered log
            # density and gradient follow from the UNTEMPERED parts kept from the last evaluation -- no pass over the data
            smc._model.bk_retemper(h._theta_dc, self._gll, self._ll_cur, t, h._grad, h._lp)
            ll = self._ll_cur
        else:
            # (logp, grad) of the current points evaluated afresh
            h._materialize(h._eval_grad(h._theta_dc, h._grad, h._lp), h._grad)
            ll = self._target.ll_last if track else None
            if tg:
                # (a copy with the STATE's row pitch: bk_retemper / select_columns walk it beside theta; clone() of a
                # padded view would come back dense)
                src = self._target.gll_last
                self._gll = torch.empty_strided(src.shape, src.stride(), dtype=src.dtype, device=src.device)
                self._gll.copy_(src)
        h._have_cache = True
        acc0 = int(h._accepted.item()) if hasattr(h, "_accepted") else 0
        for _ in range(self.draws):
            h._run_draw(h._draw)
            h._draws += 1
            if track and self.steps > 0:
                ll = torch.where(h._mask.bool(), self._target.ll_last, ll)   # accepted particles: the end point's
                if tg:
                    smc._ops.select_columns(h._mask, self._gll, self._target.gll_last)
        self.accept_rates.append((int(h._accepted.item()) - acc0) / max(1, self.draws * smc.M))
        self._ll = ll
        self._ll_cur, self._parts_ok = ll, False   # (valid again once the resampling has been applied to them)
        smc._theta_dc.copy_(h._theta_dc)

    def resampled(self, smc, idx):
        """The SMC has replaced particle m by particle idx[m] (single rank): the tracked parts follow."""
        if self._gll is None or self._ll_cur is None:
            return
        new = torch.empty_like(self._gll)
        smc._ops.gather_columns(idx, self._gll, new)
        self._gll = new
        self._ll_cur = self._ll_cur[idx.to(torch.int64)].contiguous()
        self._parts_ok = True

    def loglik(self, smc):
        """Untempered log likelihood of the particles as the move left them (None: not tracked)."""
        ll, self._ll = self._ll, None
        return ll


def hmc_kernel(stepsize: float, steps: int, draws: int = 1, metric_diag=None, metric_dense=None,
               adapt_metric: bool = False) -> _HMCKernel:
    return _HMCKernel(stepsize, steps, draws, metric_diag, metric_dense, adapt_metric)


def metropolis_kernel(scale: float) -> _RWMKernel:
    return _RWMKernel(scale)


def mala_kernel(epsilon: float, steps: int = 1) -> _MALAKernel:
    return _MALAKernel(epsilon, steps)


class TemperedLikelihoodSMC:
    MIN_ADAPTIVE_STEP = 1e-12  # the adaptive ladder never advances by less (a degenerate weight set warns and takes this step)

    def __init__(self, model, M: int, N: int, sample_initial, kernel, *, seed=None, slot_id0: int = 0,
                 group=None, ops=None, adaptive=None, max_steps: int = 100000):
        """M = particles held by THIS rank.  Across ranks (one process per GPU) pass the rank's
        first global slot as slot_id0; resampling is then global: weights and particles are
        all-gathered (RCCL over xGMI), every rank builds the same cumulative weights and draws
        its own slots' indices into the global population.

        adaptive (EXTENSION, not in the reference): a fraction in (0, 1).  The reference's ladder is t_n = n / N
        (smc.py:42-43); on a large data set (config 5: 10^6 observations, log likelihoods differing by 10^3-10^4
        between prior draws) any affordable N lets the first reweighting collapse the particle system onto one or two
        particles.  With adaptive = a the next temperature is instead the largest t <= 1 at which the effective
        sample size of the incremental weights exp((t - t_prev) * loglik) is still a * (number of particles, all
        ranks): a geometric-looking ladder of as many steps as the problem needs; N is then ignored, run() ends
        when t reaches 1, and `temperatures` / `ess_history` record the ladder."""
        self.M, self.N = int(M), int(N)
        if adaptive is not None and not (0.0 < float(adaptive) < 1.0):
            raise ValueError("adaptive must be a fraction in (0, 1): the effective sample size kept by every reweighting")
        self.adaptive = None if adaptive is None else float(adaptive)
        self.max_steps = int(max_steps)
        self.t = 0.0                 # temperature reached so far
        self.temperatures = []       # after every transition
        self.ess_history = []        # effective sample size of every reweighting (all ranks' particles)
        self._group = group
        self._slot0 = int(slot_id0)
        self._model = model
        # A reference-style model scores ONE particle per call on the host (smc.py:28-32); a
        # batched device model (batched = True) scores all of them at once.
        self._batched = getattr(model, "batched", False) is True
        self.kernel = kernel
        self._ops = ops if ops is not None else _lib.default_ops()
        dev = self._ops.device
        # initial particles: a callable i -> array-like (smc.py:24) or an (M, D) array / tensor
        if callable(sample_initial):
            init = np.array([np.asarray(sample_initial(i), dtype=np.float64).reshape(-1) for i in range(self.M)])
        else:
            init = sample_initial
        init_t = torch.as_tensor(init, dtype=torch.float64)
        if init_t.dim() != 2 or init_t.shape[0] != self.M:
            raise ValueError(f"initial particles must have shape ({self.M}, D)")
        self.D = int(init_t.shape[1])
        f64 = dict(dtype=torch.float64, device=dev)
        self._theta_dc = init_t.t().contiguous().to(dev)
        self._prop_dc = torch.empty_like(self._theta_dc)
        # reference-stream mode: the randomness comes from numpy's legacy stream, in the reference's order
        self._ref_stream = None
        if seed is np.random or isinstance(seed, np.random.RandomState) or (
                seed is not None and all(hasattr(seed, a) for a in ("standard_normal", "uniform", "random_sample"))
                and not isinstance(seed, (np.random.Generator, np.random.BitGenerator))):
            if not isinstance(kernel, _RWMKernel):
                raise ValueError("the reference stream (seed=np.random / a RandomState) reproduces the reference's run, "
                                 "whose only move kernel is metropolis_kernel(scale)")
            if adaptive is not None or self._slot0 != 0:
                raise ValueError("the reference stream is one sequential stream: t = n / N ladder, one rank")
            self._ref_stream = seed
            self._rng_kind = self._rng_state = None
        else:
            self._rng_kind, self._rng_state = make_streams(seed, self.M, self._slot0, False, dev)
        self._logu = torch.empty(self.M, **f64)
        self._u = torch.empty(self.M, **f64)
        self._cdf = torch.empty(self.M, **f64)
        self._idx = torch.empty(self.M, dtype=torch.int32, device=dev)
        self._mask = torch.empty(self.M, dtype=torch.uint8, device=dev)
        self._accepted = torch.zeros(1, dtype=torch.int32, device=dev)
        self.last_ess = float("nan")

    # -- reference API ------------------------------------------------------------------------
    @property
    def thetas(self):
        """(M, D) particles: a device view for batched models, a NumPy array (as in the reference)
        for reference-style host models."""
        if self._batched:
            return self._theta_dc.t()
        return np.array(self._theta_dc.t().cpu().numpy())

    def _per_particle(self, fn, Theta):
        th = np.array(Theta.cpu().numpy())
        vals = [float(fn(th[m])) for m in range(th.shape[0])]
        return torch.tensor(vals, dtype=torch.float64, device=self._ops.device)

    def log_prior(self, Theta):
        if self._batched:
            return self._model.log_prior(Theta)
        return self._per_particle(self._model.log_prior, Theta)

    def log_likelihood(self, Theta):
        if self._batched:
            return self._model.log_likelihood(Theta)
        return self._per_particle(self._model.log_likelihood, Theta)

    def time(self, n: int) -> float:
        if self.adaptive is not None:
            # the ladder as far as it has been built (n = 0: the prior)
            return 0.0 if n <= 0 else self.temperatures[min(n, len(self.temperatures)) - 1]
        return n / self.N

    def run(self) -> None:
        if self.adaptive is not None:
            n = len(self.temperatures)
            while self.t < 1.0:
                n += 1
                if n > self.max_steps:
                    raise RuntimeError(f"adaptive ladder did not reach t = 1 in {self.max_steps} steps (t = {self.t})")
                self.transition(n)
            return
        for n in range(1, self.N + 1):
            self.transition(n)

    def __iter__(self):
        self.run()
        return iter(self.thetas)

    # -- one temperature step ---------------------------------------------------------------------
    def _tempered(self, theta_dc, t: float):
        """log_likelihood * t + log_prior for every particle [smc.py:47-51]."""
        Th = theta_dc.t()
        return (self.log_likelihood(Th) * t + self.log_prior(Th)).contiguous()

    def _multi(self):
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1

    def _next_temperature(self, ll):
        """Largest step d <= 1 - t with ESS(exp(d * ll)) >= adaptive * (particles of all ranks): 64-point searches
        on the device (the ESS falls monotonically in d), every rank on the same gathered vector."""
        if self._multi():
            from . import dist as _bkdist
            from .diagnostics import _all_gather_counts

            counts = _all_gather_counts(ll.shape[0], ll.device, self._group)
            mmax = max(counts)
            pad = ll if ll.shape[0] == mmax else torch.cat([ll, ll.new_full((mmax - ll.shape[0],), float("-inf"))])
            ll = torch.cat([p[:c] for p, c in zip(_bkdist.all_gather(pad.contiguous(), self._group), counts)])
        # -inf is a legitimate zero-weight particle (a hard constraint: exp(d * -inf) = 0 in the ESS and in the
        # reweighting).  NaN / +inf make every ESS comparison below False: the ladder would creep forward by 2^-64 of
        # the remaining way per temperature, a full move each, until max_steps -- as would a system with no finite
        # particle left
        broken = torch.isnan(ll) | (ll == float("inf"))
        if bool(broken.any().item()) or not bool(torch.isfinite(ll).any().item()):
            bad = int(broken.sum().item())
            raise FloatingPointError(f"adaptive SMC ladder at t = {self.t}: {bad} of {ll.numel()} particles have a NaN / +inf "
                                     f"log likelihood ({int((ll == float('-inf')).sum().item())} are -inf); the next "
                                     "temperature cannot be chosen")
        x = ll - ll.max()
        want = self.adaptive * x.shape[0]

        def ess(d):  # d: [k] candidates -> [k]
            w = torch.exp(d[:, None] * x[None, :])
            return w.sum(dim=1) ** 2 / (w * w).sum(dim=1)

        room = 1.0 - self.t
        full = torch.tensor([room], dtype=torch.float64, device=x.device)
        if float(ess(full)[0].item()) >= want:
            return room
        # a geometric bracket first (the first step of a large data set is 1e-4 .. 1e-6 of the way), then two linear
        # refinements: the step is found to 1 / 4096 of itself, three device round trips in all
        halves = room * torch.pow(torch.tensor(0.5, dtype=torch.float64, device=x.device),
                                  torch.arange(1, 65, dtype=torch.float64, device=x.device))
        ok = ess(halves) >= want
        if not bool(ok.any().item()):
            import warnings

            warnings.warn(f"adaptive SMC ladder at t = {self.t}: no step down to 2^-64 of the remaining way keeps the ESS at "
                          f"{self.adaptive} of the particles; taking the minimum step {self.MIN_ADAPTIVE_STEP:g}", stacklevel=2)
            return min(room, self.MIN_ADAPTIVE_STEP)
        k = int(ok.to(torch.int64).argmax().item())
        lo, hi = float(halves[k].item()), float(halves[k].item()) * 2.0  # (nothing passes: the smallest candidate; the ESS record shows it)
        for _ in range(2):
            cand = lo + (hi - lo) * torch.arange(1, 65, dtype=torch.float64, device=x.device) / 64.0
            passed = int((ess(cand) >= want).to(torch.int64).sum().item())  # (monotone: the first `passed` candidates pass)
            lo, hi = (lo + (hi - lo) * passed / 64.0), (lo + (hi - lo) * min(64, passed + 1) / 64.0)
            if passed == 64:
                break
        return lo

    def transition(self, n: int) -> None:
        ops = self._ops
        t_prev = self.t if self.adaptive is not None else self.time(n - 1)
        self.kernel.move(self, t_prev)                                # smc.py:53-57
        th = self._theta_dc
        Th = th.t()
        # one evaluation of the two parts serves both densities of smc.py:47-51 (the reference evaluates each twice);
        # a move kernel that tracked the particles' log likelihood hands it over
        ll = self.kernel.loglik(self) if hasattr(self.kernel, "loglik") else None
        if ll is None:
            ll = self.log_likelihood(Th)
        if self.adaptive is not None:
            t_next = min(1.0, self.t + self._next_temperature(ll))
            if 1.0 - t_next < 1e-12:
                t_next = 1.0
            logw = ll * (t_next - t_prev)
        else:
            t_next = self.time(n)
            lprior = self.log_prior(Th)
            lpm1 = (ll * t_prev + lprior).contiguous()
            lp = (ll * t_next + lprior).contiguous()
            logw = lp - lpm1
        import torch.distributed as dist

        multi = self._multi()
        # weights exp(lp - lpminus1) [smc.py:67-70], scaled by exp(-max) so that a large data set
        # (log-likelihood steps of 1e4 and more) cannot underflow every weight to zero; the
        # reference normalises by the sum (smc.py:73), so the common factor changes nothing
        if self._ref_stream is not None:
            if multi:
                raise RuntimeError("the reference stream is one sequential stream: one rank")
            w = torch.exp(logw).contiguous()                      # smc.py:67-70 as written
            self._u.copy_(torch.as_tensor(np.asarray(self._ref_stream.random_sample(self.M), dtype=np.float64)))
        else:
            top = logw.max()
            if multi:
                dist.all_reduce(top, op=dist.ReduceOp.MAX, group=self._group)
            w = torch.exp(logw - top).contiguous()
            ops.uniform(self._rng_kind, self._rng_state, self._u)
        if multi:
            self._resample_across_ranks(w, th)
        else:
            tot = float(w.sum().item())
            if not (tot > 0.0 and math.isfinite(tot)):
                # (the reference: np.random.choice raises "probabilities contain NaN" / "do not sum to 1" here)
                raise FloatingPointError(f"SMC reweighting at t = {t_next}: the weights sum to {tot} (every particle has zero "
                                         "weight, or a log density is NaN / +inf): nothing to resample from")
            self.last_ess = float((tot ** 2 / (w * w).sum()).item())
            ops.resample_indices(w, self._u, self._cdf, self._idx)   # smc.py:73: choice(p = w / w.sum()), its arithmetic
            ops.gather_columns(self._idx, th, self._prop_dc)         # thetas[idxs], smc.py:75
            if hasattr(self.kernel, "resampled"):
                self.kernel.resampled(self, self._idx)
        self._theta_dc, self._prop_dc = self._prop_dc, self._theta_dc
        self.t = t_next
        self.temperatures.append(t_next)
        self.ess_history.append(self.last_ess)

    def _resample_across_ranks(self, w, th):
        """Multinomial resampling over the particles of ALL ranks (smc.py:64-75 with the particle
        array sharded): the weights are all-gathered (8 B per particle), every rank draws the
        global ancestor index of each of its own slots, and the ancestors' columns travel
        point to point -- one all_to_all of requests (indices) and one of particles, M_local*D*8
        bytes received per rank instead of the M_total*D*8 an all_gather of particles costs.
        On a node this is RCCL over the direct xGMI links."""
        import torch.distributed as dist

        ops, g = self._ops, self._group
        from . import dist as _bkdist
        from .diagnostics import _all_gather_counts

        world, M = dist.get_world_size(g), w.shape[0]
        # particle counts may differ by one between ranks (dist.shard gives the remainder to the
        # first ranks): pad the weight shards to the widest for the collective, and find an
        # ancestor's owner from the cumulative counts rather than by division
        counts = _all_gather_counts(M, w.device, g)
        mmax = max(counts)
        send_w = w if M == mmax else torch.cat([w, torch.zeros(mmax - M, dtype=w.dtype, device=w.device)])
        wparts = _bkdist.all_gather(send_w, g)
        w_all = torch.cat([p[:c] for p, c in zip(wparts, counts)])
        self.last_ess = float((w_all.sum() ** 2 / (w_all * w_all).sum()).item())
        ops.resample_indices(w_all, self._u, torch.empty_like(w_all), self._idx)  # global ancestors
        idx = self._idx.to(torch.int64)
        ends = torch.cumsum(torch.tensor(counts, dtype=torch.int64, device=idx.device), 0)
        owner = torch.searchsorted(ends, idx, right=True)     # rank whose block holds global particle idx
        first = ends - torch.tensor(counts, dtype=torch.int64, device=idx.device)
        # my slots grouped by the rank that owns their ancestor (stable: bk_sort_by_key)
        _, order = ops.sort_by_key(owner.to(torch.float64), torch.arange(owner.numel(), dtype=torch.int64, device=idx.device))
        want = torch.bincount(owner, minlength=world)         # how many columns I need from each rank
        give = torch.empty_like(want)
        _bkdist.all_to_all_single(give, want, group=g)         # how many each rank needs from me
        want_l, give_l = want.tolist(), give.tolist()
        req_out = (idx[order] - first[owner[order]]).contiguous()
        req_in = torch.empty(sum(give_l), dtype=torch.int64, device=idx.device)
        _bkdist.all_to_all_single(req_in, req_out, give_l, want_l, group=g)
        # the requested columns, particle-major so that each destination's block is contiguous
        D, n_send = th.shape[0], req_in.shape[0]
        send_pm = torch.empty((n_send, D), dtype=th.dtype, device=th.device)
        if n_send:
            cols = torch.empty((D, n_send), dtype=th.dtype, device=th.device)
            ops.gather_columns(req_in.to(self._idx.dtype), th, cols)
            ops.relayout(cols, send_pm.t())
        recv_pm = torch.empty((M, D), dtype=th.dtype, device=th.device)
        _bkdist.all_to_all_single(recv_pm, send_pm, want_l, give_l, group=g)
        # arrival k belongs to slot order[k]
        self._prop_dc[:, order] = recv_pm.t()


class TorchPriorLikelihoodModel:
    """Batched ``LogPriorLikelihoodModel`` from two PyTorch functions of a (M, D) tensor."""

    batched = True

    def __init__(self, log_prior, log_likelihood, dims: int):
        self._lp, self._ll, self._D = log_prior, log_likelihood, int(dims)

    def dims(self) -> int:
        return self._D

    def log_prior(self, Theta):
        with torch.no_grad():
            return self._lp(Theta)

    def log_likelihood(self, Theta):
        with torch.no_grad():
            return self._ll(Theta)

    def log_density(self, Theta):
        return self.log_prior(Theta) + self.log_likelihood(Theta)

    def log_density_gradient_tempered(self, Theta, t: float):
        x = Theta.detach().requires_grad_(True)
        with torch.enable_grad():
            lp = self._ll(x) * t + self._lp(x)
            (g,) = torch.autograd.grad(lp.sum(), x)
        return lp.detach(), g

    def log_density_gradient(self, Theta):
        return self.log_density_gradient_tempered(Theta, 1.0)
