"""Metropolis / Metropolis-Hastings for many chains: drop-in for ``bayes_kit/metropolis.py``.

The two accept rules (``metropolis.py:12-38`` and ``:41-76``) are the ``bk_mh_accept`` kernel
(strict ``<`` against ``log(uniform)`` from each chain's own stream); MALA, the SMC move
kernels and the classes below all go through it.  The sampler classes keep the reference's
constructor signatures (``metropolis.py:80-88``, ``:139-145``) and ``sample()`` protocol.
The proposal and its transition density are user callbacks in the reference; here they are
called once per draw for ALL chains:

    proposal_fn(Theta) -> (C, D) tensor          Theta: (C, D) fp64 device view, strides (1, ld)
    transition_lp_fn(To, From) -> (C,) tensor    log q(To | From), one value per chain

With a reference-style single-chain NumPy model the callbacks get and return NumPy vectors
exactly as in the reference (one chain; the accept still runs on the device from the
uploaded generator state, so a seeded run reproduces the reference's draws).

``ChainRng`` hands user proposals the same counter-based per-chain streams the samplers use.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from ._engine import ManyChainSampler, make_streams, numpy_generator_from_words


class ChainRng:
    """Per-chain device random streams for user code (proposal callbacks).

    Chain c draws from ``numpy.random.Philox(key=[seed, chain_id0 + c])`` -- bit for bit what
    ``np.random.Generator(Philox(key=...))`` yields, so a many-chain proposal can be checked
    against a loop of single-chain NumPy proposals.
    """

    def __init__(self, seed: int, chains: int, chain_id0: int = 0, ops=None):
        self._ops = ops if ops is not None else _lib.default_ops()
        self._C = int(chains)
        self._kind, self._state = make_streams(int(seed), self._C, int(chain_id0), False, self._ops.device)

    def normal(self, Loc: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """``Loc + scale * z`` with z ~ N(0, I): (C, D) in, (C, D) out (`rng.normal(loc, scale)`)."""
        C, D = Loc.shape
        if C != self._C:
            raise ValueError(f"this ChainRng serves {self._C} chains, got {C}")
        loc_dc = Loc.t()
        if loc_dc.stride(1) != 1:
            tmp = torch.empty((D, C), dtype=torch.float64, device=self._ops.device)
            self._ops.relayout(loc_dc, tmp)
            loc_dc = tmp
        out = torch.empty((D, C), dtype=torch.float64, device=self._ops.device)
        self._ops.momentum_refresh(self._kind, self._state, loc_dc, 1.0, float(scale), out, None, None)
        return out.t()

    def uniform(self) -> torch.Tensor:
        """One U[0, 1) per chain (`rng.uniform()`)."""
        u = torch.empty(self._C, dtype=torch.float64, device=self._ops.device)
        self._ops.uniform(self._kind, self._state, u)
        return u

    def generator(self, chain: int = 0) -> np.random.Generator:
        """A NumPy Generator positioned where `chain`'s stream is now (a copy)."""
        w = self._state.cpu().numpy().view(np.uint64)
        return numpy_generator_from_words(self._kind, w[:, chain])


def _accept(ops, rng_kind, rng_state, lp_proposal, lp_current, fwd, rev):
    C = lp_proposal.shape[0]
    dev = ops.device
    logu = torch.empty(C, dtype=torch.float64, device=dev)
    mask = torch.empty(C, dtype=torch.uint8, device=dev)
    ops.log_uniform(rng_kind, rng_state, logu)
    # lp_cur is updated in place by the kernel on acceptance: work on a copy, callers own theirs
    ops.mh_accept(_lib.ACCEPT_MALA, lp_current.clone(), fwd, lp_proposal, rev, logu, mask, None, None)
    return mask.bool()


def _as_lp(x, C, dev):
    t = torch.as_tensor(x, dtype=torch.float64, device=dev).reshape(-1)
    if t.shape[0] != C:
        raise ValueError(f"expected {C} log densities, got {t.shape[0]}")
    return t.contiguous()


def _accept_host_rng(lp_proposal, lp_current, fwd, rev, rng) -> bool:
    """The reference's scalar form: ``rng`` is a host generator (anything with ``uniform()``, e.g. a
    ``numpy.random.Generator``), the log densities are host floats.  In the reference this is a pure
    scalar helper (metropolis.py:12-76), so it stays one here: one uniform is taken from ``rng`` and
    the comparison is evaluated on the host in the reference's operation order -- no device round
    trip for a single comparison.  The many-chain form (``ChainRng``) is the ``bk_mh_accept`` kernel."""
    with np.errstate(divide="ignore"):
        log_u = np.log(rng.uniform())
    if fwd is None and rev is None:
        return bool(log_u < lp_proposal - lp_current)  # metropolis.py:37-38
    return bool(log_u < (lp_proposal - lp_current) + (rev - fwd))  # metropolis.py:70-76


def metropolis_accept_test(lp_proposal, lp_current, rng):
    """``log(rng.uniform()) < lp_proposal - lp_current`` (metropolis.py:12-38).  With a ``ChainRng``:
    (C,) tensors in, (C,) bool tensor out, one uniform of every chain's stream consumed.  With a
    host generator and floats: a bool, as in the reference."""
    if not isinstance(rng, ChainRng):
        return _accept_host_rng(lp_proposal, lp_current, None, None, rng)
    dev = rng._ops.device
    lp_p, lp_c = _as_lp(lp_proposal, rng._C, dev), _as_lp(lp_current, rng._C, dev)
    return _accept(rng._ops, rng._kind, rng._state, lp_p, lp_c, None, None)


def metropolis_hastings_accept_test(lp_proposal, lp_current, lp_forward_transition, lp_reverse_transition, rng):
    """``log(rng.uniform()) < (lp_proposal - lp_current) + (lp_reverse - lp_forward)``
    (metropolis.py:41-76); many-chain or scalar form as above."""
    if not isinstance(rng, ChainRng):
        return _accept_host_rng(lp_proposal, lp_current, lp_forward_transition, lp_reverse_transition, rng)
    dev, C = rng._ops.device, rng._C
    return _accept(rng._ops, rng._kind, rng._state, _as_lp(lp_proposal, C, dev), _as_lp(lp_current, C, dev),
                   _as_lp(lp_forward_transition, C, dev), _as_lp(lp_reverse_transition, C, dev))


class MetropolisHastings(ManyChainSampler):
    """metropolis.py:79-135."""

    def __init__(self, model, proposal_fn: Callable, transition_lp_fn: Callable, *, init=None, seed=None,
                 chains: Optional[int] = None, chain_id0: int = 0, ops=None):
        self._proposal_fn = proposal_fn
        self._transition_lp_fn = transition_lp_fn
        self._setup(model, None, init, seed, chains, chain_id0, ops)
        self._init_graph(False)
        D, C, dev = self._dim, self._C, self._ops.device
        f64 = dict(dtype=torch.float64, device=dev)
        self._theta_p = torch.empty((D, C), **f64)
        self._lp = torch.empty(C, **f64)
        self._lp_p = torch.empty(C, **f64)
        self._fwd = torch.empty(C, **f64)
        self._rev = torch.empty(C, **f64)
        self._logu = torch.empty(C, **f64)
        self._mask = torch.empty(C, dtype=torch.uint8, device=dev)
        self._accepted = torch.zeros(1, dtype=torch.int32, device=dev)
        self._draws = 0
        self._eval_logp(self._theta_dc, self._lp)  # metropolis.py:99

    def _state_tensors(self):
        return {"theta": self._theta_dc, "lp": self._lp, "accepted": self._accepted}

    @property
    def _log_p_theta(self):
        return self._lp if self._batched else float(self._lp[0].item())

    @property
    def last_accept(self):
        return self._mask.bool() if self._batched else bool(self._mask[0].item())

    def accept_rate(self) -> float:
        n = self._draws * self._C
        return float(self._accepted.item()) / n if n else float("nan")

    def __next__(self):
        return self.sample()

    # -- user callbacks ---------------------------------------------------------------------------
    def _propose(self):
        """theta* = proposal_fn(theta) into the proposal buffer (metropolis.py:113-119)."""
        D, C = self._dim, self._C
        if self._batched:
            prop = self._proposal_fn(self._theta_dc.t())
            prop = torch.as_tensor(prop).to(device=self._ops.device, dtype=torch.float64)
            if tuple(prop.shape) != (C, D):
                raise ValueError(f"proposal_fn must return a ({C}, {D}) array, got {tuple(prop.shape)}")
            self._ops.relayout(prop.t(), self._theta_p)
        else:
            # np.asanyarray(..., dtype=float64) raises ValueError for non-numeric proposals, as in
            # the reference (metropolis.py:114-116, test_metropolis.py:289-295)
            prop = np.asanyarray(self._proposal_fn(np.array(self._theta_dc[:, 0].cpu().numpy())), dtype=np.float64)
            self._theta_p[:, 0].copy_(torch.from_numpy(np.array(np.broadcast_to(prop, (D,)))))

    def _transition_lps(self):
        """(forward, reverse) = (q(theta* | theta), q(theta | theta*))   metropolis.py:126-127."""
        th, thp = self._theta_dc, self._theta_p
        if self._batched:
            self._fwd.copy_(torch.as_tensor(self._transition_lp_fn(thp.t(), th.t())).reshape(-1))
            self._rev.copy_(torch.as_tensor(self._transition_lp_fn(th.t(), thp.t())).reshape(-1))
        else:
            a, b = np.array(th[:, 0].cpu().numpy()), np.array(thp[:, 0].cpu().numpy())
            self._fwd.fill_(float(self._transition_lp_fn(b, a)))
            self._rev.fill_(float(self._transition_lp_fn(a, b)))
        return self._fwd, self._rev

    # -- one draw for every chain --------------------------------------------------------------------
    def sample(self):
        ops = self._ops
        self._propose()
        self._eval_logp(self._theta_p, self._lp_p)
        fwd, rev = self._transition_lps()
        ops.log_uniform(self._rng_kind, self._rng_state, self._logu)
        # on acceptance the kernel also moves lp_prop into lp (metropolis.py:121-123)
        ops.mh_accept(_lib.ACCEPT_MALA, self._lp, fwd, self._lp_p, rev, self._logu, self._mask, None,
                      self._accepted)
        self._select(self._mask, self._theta_dc, self._theta_p)
        self._draws += 1
        return self._draw_out(self._theta_dc, self._lp)


class Metropolis(MetropolisHastings):
    """metropolis.py:138-155: symmetric proposals, no transition densities."""

    def __init__(self, model, proposal_fn: Callable, *, init=None, seed=None, chains: Optional[int] = None,
                 chain_id0: int = 0, ops=None):
        super().__init__(model, proposal_fn, None, init=init, seed=seed, chains=chains, chain_id0=chain_id0,
                         ops=ops)

    def _transition_lps(self):
        return None, None
