"""Many-chain delayed-rejection generalized HMC on the GPU: drop-in for
``bayes_kit/drghmc.py:10-446`` (Modi, Barnett & Carpenter 2023).

Same constructor, validation (error types and texts), ``sample()`` and iterator protocol
as the reference's ``DrGhmcDiag``.  The reference evaluates the acceptance probability of
proposal k recursively, making "ghost" proposals from the proposal itself and keeping
gradients on a stack (drghmc.py:82, 391-446).  Here every phase-space point carries its
own (logp, grad) and the recursion runs in lockstep over SETS of chains:

* level 0 holds the proposals P_k of the chains still inside the stage loop, level 1 the
  ghosts G_i(P_k), level 2 the ghosts of ghosts, ... (at most max_proposals levels);
* before each trajectory the chains that need it are compacted into a dense [D, n]
  buffer (``bk_compact_indices`` + the gathering first leapfrog step), so a rarely needed
  160-step stage costs bandwidth only for the few chains that reach it;
* per-chain scalars (joint log density H, Hastings term h, log acceptance a) are updated
  by the ``bk_dr_*`` kernels; uniforms come from each chain's own stream in the reference's
  order (retry uniform, then accept uniform, per attempted stage; drghmc.py:370,378).

The momentum flip at the end of a draw (drghmc.py:388) is folded into the next partial
refresh: ``(-rho)*sqrt(1-damping)`` and ``rho*(-sqrt(1-damping))`` are the same double.
"""
from __future__ import annotations

import math
from collections.abc import Sequence
from typing import Optional

import numpy as np
import torch

from ._engine import ManyChainSampler


class _Level:
    """Dense buffers of one recursion level."""

    def __init__(self, D, C, dev, pad=0):
        f64 = dict(dtype=torch.float64, device=dev)
        self.theta = torch.empty((D, C + pad), **f64)[:, :C]
        self.rho = torch.empty((D, C + pad), **f64)[:, :C]
        self.grad = torch.empty((D, C + pad), **f64)[:, :C]
        self.logp = torch.empty(C, **f64)
        self.kin = torch.empty(C, **f64)
        self.H = torch.empty(C, **f64)
        self.h = torch.empty(C, **f64)
        self.a = torch.empty(C, **f64)
        self.live = torch.empty(C, dtype=torch.uint8, device=dev)
        self.idx = torch.empty(C, dtype=torch.int32, device=dev)
        self.idx_alt = torch.empty(C, dtype=torch.int32, device=dev)  # (a list is built while the previous one is read)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.accepted = torch.empty(C, dtype=torch.uint8, device=dev)


class DrGhmcDiag(ManyChainSampler):
    TUNING = ("graph", "device_counts", "fuse_first_ghost", "recompute_gradient", "tune_placement")

    def __init__(
        self,
        model,
        max_proposals: int,
        leapfrog_step_sizes: Sequence[float],
        leapfrog_step_counts: Sequence[int],
        damping: float,
        metric_diag=None,
        init=None,
        seed=None,
        prob_retry: bool = True,
        *,
        chains: Optional[int] = None,
        chain_id0: int = 0,
        path: str = "auto",
        tuning: Optional[dict] = None,
        ops=None,
        **knobs,
    ):
        """The reference's arguments (drghmc.py:37-48), then the engine's (ManyChainSampler: chains, chain_id0, path, tuning,
        ops).  Tuning knobs (none changes a result): ``graph`` (replay a draw as one hipGraph; default on where the draw is a
        fixed launch sequence), ``device_counts`` (lane-set sizes stay on the device; default on where the model allows),
        ``fuse_first_ghost`` (the first ghost proposal inside the stage's launch), ``recompute_gradient`` (one-launch path:
        no gradient cache -- every proposal launch evaluates its source point's gradient itself, two arrays instead of three
        per launch and per scatter), ``tune_placement``."""
        fuse_builtin, fuse_steps = self._resolve_path(path)
        tn = self._resolve_tuning(tuning, knobs)
        tune_placement, device_counts, graph = tn.get("tune_placement"), tn.get("device_counts"), tn.get("graph")
        fuse_first_ghost = tn.get("fuse_first_ghost", True)
        self._max_proposals = max_proposals
        self._leapfrog_step_sizes = leapfrog_step_sizes
        self._leapfrog_step_counts = leapfrog_step_counts
        self._damping = damping
        self._prob_retry = prob_retry
        # the reference validates after building its state (drghmc.py:83); nothing of that
        # state is observable when validation fails, so validate first and allocate after
        self._validate_arguments()
        self._setup(model, metric_diag, init, seed, chains, chain_id0, ops)
        D, C, dev = self._dim, self._C, self._ops.device
        f64 = dict(dtype=torch.float64, device=dev)
        # rho0 = rng.normal(size=D) AFTER theta0 (drghmc.py:77)
        self._rho_dc = torch.empty((D, C), **f64)
        self._ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, self._rho_dc, None, None,
                                   None, self._rng_work)
        self._rho_sign = 1.0  # stored momentum is sign * (reference's _rho)
        self._grad = torch.empty((D, C), **f64)
        self._lp = torch.empty(C, **f64)
        self._kin = torch.empty(C, **f64)
        self._cur_H = torch.empty(C, **f64)
        self._cur_h = torch.empty(C, **f64)
        self._rej = torch.empty(C, **f64)
        self._alive = torch.empty(C, dtype=torch.uint8, device=dev)
        self._levels = None      # (allocated below, once the path is known)
        self._level0_alt = None  # a second level-0 buffer set (device-count path: stages alternate, see _draw_dev)
        self._have_cache = False
        self._draws = 0
        self._draws_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # the draw count, kept on the device as well
        self._attached = []  # (RunningMoments / DrawRecorder objects fed by the draw itself: attach())
        # built-in targets that can run a whole proposal (gather, L_k steps, flip, energies) in
        # one launch with the gradient inlined (bk_dr_proposal_funnel); same results
        self._fused = bool(fuse_builtin) and self._batched and hasattr(model, "bk_dr_proposal")
        # ... and models whose leapfrog step {gradient, kick, drift} is one launch (bk_leapfrog_step: a lane-spread density
        # compiled from source): the step-by-step paths then issue one launch per step instead of two; same results
        self._step_hook = bool(fuse_steps) and self._batched and hasattr(model, "bk_leapfrog_step")
        # ... and models whose whole trajectory {gathering first step .. last gradient + log density} is one launch
        # (bk_leapfrog_trajectory: a per-chain density compiled from source): 2-3 launches per proposal; same results
        self._traj_hook = bool(fuse_builtin) and self._step_hook and hasattr(model, "bk_leapfrog_trajectory")
        # Lane counts on the device: the sizes of the lane sets -- which depend on the draw's own accept / retry
        # decisions -- stay in device memory; every launch is sized for its parent set and surplus workgroups
        # exit at once.  No host read inside sample(), so the draw is a FIXED launch sequence and replays as one
        # hipGraph.  Two kinds of model can run that way:
        #   * one whose whole proposal is one library launch (bk.Funnel with the gradient inlined): _one_launch;
        #   * ANY model whose gradient op takes its chain count from device memory (`bk_counted`: the built-in
        #     targets' bk_target_*_grad_n, a CTarget with a counted_symbol of type bk_target_fn_n) -- the
        #     trajectory is then the reference's own sequence of model calls (drghmc.py:280-283), one counted
        #     gradient launch + one counted kick+drift launch per leapfrog step.
        # Other models (PyTorch autograd: the call needs a host-side shape) keep compacted launches sized by
        # three host reads per draw.
        one_launch = self._fused and getattr(model, "bk_dr_proposal_supported", lambda: False)()
        counted = self._batched and hasattr(model, "bk_eval") and getattr(model, "bk_counted", False) is True
        can_dev = one_launch or counted
        if device_counts and not can_dev:
            raise ValueError("device_counts=True needs a model whose gradient op takes its chain count from device "
                             "memory (a built-in target, or a CTarget with counted_symbol=)")
        if device_counts is None:
            # a counted step-by-step draw is ~2 launches per leapfrog step over every lane set: it wins where the
            # draw is launch-bound; HBM-bound shapes keep the compacted 16-byte-per-lane kernels of the host-sized path
            device_counts = one_launch or (counted and dev.type == "cuda" and D * C <= self.GRAPH_AUTO_MAX_ELEMS)
        self._dev_counts = bool(device_counts)
        self._one_launch = self._dev_counts and one_launch
        if graph is None:
            graph = self._dev_counts and dev.type == "cuda"
        if graph and not self._dev_counts:
            raise ValueError("graph=True needs device_counts (the host-sized path reads lane counts back every draw)")
        # (one-launch path) a proposal's launch also runs the first ghost of the lanes it produces; False keeps
        # that ghost a launch of its own -- same results, for A/B timing and the tests
        self._fuse_first_ghost = bool(fuse_first_ghost) and self._one_launch
        # (one-launch path) no gradient cache: a proposal launch evaluates the gradient of its source point itself (the same
        # values: a function of theta alone, sums in one order) instead of reading one that an earlier launch stored and a
        # scatter moved -- the launches over all chains are memory launches, and this takes a third of their bytes away.
        # self._grad is then NOT kept up to date between draws; state_dict() refreshes it.
        self._regrad = bool(tn.get("recompute_gradient", True)) and self._one_launch
        self._init_graph(graph)
        self._graph_many = {}  # advance(n): graphs of several consecutive draws, by their number
        self.host_syncs_per_draw = 0 if self._dev_counts else max(0, int(max_proposals) - 1) + sum(
            max(0, k - 1) for k in range(int(max_proposals)))
        # Row padding (one-launch-proposal path).  The proposal kernel walks the ROWS of a few chains: with a
        # row pitch that is a multiple of 4 KiB -- 32,768 chains: 256 KiB -- every row of a chain sits on the same
        # memory channel, and the gather / store phase of the first stage (all chains, 160 MB) queues up there:
        # 48 -> 34 us with the rows 576 bytes further apart (tools/funnel_traj_bench.py, PAD=72).
        pad = 72 if (self._one_launch and C >= 4096 and (C * 8) % 4096 == 0) else 0
        if pad:
            def padded(t):
                p = torch.empty((D, C + pad), **f64)[:, :C]
                p.copy_(t)
                return p
            self._theta_dc, self._rho_dc, self._grad = padded(self._theta_dc), padded(self._rho_dc), padded(self._grad)
        self._levels = [_Level(D, C, dev, pad) for _ in range(int(max_proposals))]
        self.placement = None
        if self._wants_placement_tuning(tune_placement) and not self._fused:
            self._tune_placement()
        self._host_stats = (0, 0, [])   # (grad evals, lane steps, [(tag, lanes)]) of the last host-sized draw
        self._first_eval = 0
        if self._dev_counts:
            self._steps_total_base = 0.0
            self._make_schedule()
            if int(max_proposals) > 1 and self._one_launch:
                self._level0_alt = _Level(D, C, dev, pad)

    def _make_schedule(self):
        """The fixed schedule of trajectories (slot order = launch order) and its lane counters, for the
        step counts as they stand (plain attributes in the reference: they may be assigned between draws)."""
        dev = self._ops.device
        self._schedule = []
        self._plan(int(self._max_proposals))
        self._schedule_counts = tuple(int(n) for n in self._leapfrog_step_counts)
        self._slot_lanes = torch.zeros(len(self._schedule), dtype=torch.int32, device=dev)
        self._slot_steps = torch.tensor([st for _, st in self._schedule], dtype=torch.float64, device=dev)
        self._slot_lanes_total = torch.zeros(len(self._schedule), dtype=torch.int64, device=dev)
        # one lane counter per list the draw builds (stage lists + ghost lists), zeroed by the draw's first launch
        K = int(self._max_proposals)
        n_lists = (K - 1) + sum(max(0, k - 1) * (2 ** 0) for k in range(K)) + 64
        self._counters = torch.zeros(min(64, n_lists), dtype=torch.int32, device=dev)

    def _plan(self, K):
        """Tags and step counts of the trajectories of one draw in launch order, e.g. for K = 3:
        P0 | P1 G0(P1) | P2 G0(P2) G1(P2) G0(G1(P2))   (SURVEY 3.3.1)."""
        def ghosts(tag, k):
            for i in range(k):
                g = "G%d(%s)" % (i, tag)
                self._schedule.append((g, int(self._leapfrog_step_counts[i])))
                ghosts(g, i)
        for k in range(K):
            self._schedule.append(("P%d" % k, int(self._leapfrog_step_counts[k])))
            ghosts("P%d" % k, k)

    # -- statistics of the last draw ----------------------------------------------------------------------
    # With device-side lane counts these READ BACK the counters (a host synchronisation); use
    # lane_steps_device / lane_steps_total for figures that must not synchronise.
    @property
    def last_stage_lanes(self):
        """(tag, lanes) of every trajectory that had at least one lane in the last draw."""
        if not self._dev_counts:
            return self._host_stats[2]
        lanes = self._slot_lanes.cpu().tolist()
        return [(tag, n) for (tag, _), n in zip(self._schedule, lanes) if n > 0]

    @property
    def last_grad_evals(self):
        """Gradient evaluations (each over a lane set) in the last draw."""
        if not self._dev_counts:
            return self._host_stats[0]
        lanes = self._slot_lanes.cpu().tolist()
        return self._first_eval + sum(st for (_, st), n in zip(self._schedule, lanes) if n > 0)

    @property
    def last_lane_steps(self):
        """Sum over the last draw's trajectories of lanes x steps (chain-steps actually run)."""
        if not self._dev_counts:
            return self._host_stats[1]
        return int(round(float(self.lane_steps_device.item())))

    @property
    def lane_steps_device(self):
        """The same as a 0-d device tensor (no host synchronisation)."""
        if not self._dev_counts:
            return torch.tensor(float(self._host_stats[1]), dtype=torch.float64, device=self._ops.device)
        return (self._slot_lanes.to(torch.float64) * self._slot_steps).sum()

    def _tune_placement(self):
        """Every recursion level streams its own (theta, rho, grad) triple through kick+drift: the
        3 K level arrays are assigned to those roles by timing (ManyChainSampler._tune_roles)."""
        ops, m = self._ops, self._metric_dev
        h = float(self._leapfrog_step_sizes[0])
        flat = [a for lv in self._levels for a in (lv.theta, lv.rho, lv.grad)]

        def cost(a):
            return sum(self._time_ms(lambda t=a[3 * i], r=a[3 * i + 1], g=a[3 * i + 2]:
                                     ops.kick_drift(t, t, r, r, g, m, h, False, 0.0, True, h))
                       for i in range(len(self._levels)))

        chosen, rep = self._tune_roles(flat, cost)
        for i, lv in enumerate(self._levels):
            lv.theta, lv.rho, lv.grad = chosen[3 * i], chosen[3 * i + 1], chosen[3 * i + 2]
        self.placement = {"kick_drift_ms_as_allocated": rep["ms_as_allocated"],
                          "kick_drift_ms_chosen": rep["ms_chosen"], "assignments_tried": rep["assignments_tried"]}

    # -- validation: same checks, error types and texts as drghmc.py:85-207 -----------------------
    def _validate_arguments(self) -> None:
        K = self._max_proposals
        if not isinstance(K, int):
            raise TypeError(f"max_proposals must be an int, not {type(K)}")
        if not (K >= 1):
            raise ValueError(f"max_proposals must be greater than or equal to 1, not {K}")
        sizes = self._leapfrog_step_sizes
        if not isinstance(sizes, Sequence):
            raise TypeError(
                f"leapfrog_step_sizes must be an instance of type sequence, but found type {type(sizes)}")
        if len(sizes) != K:
            raise ValueError(
                f"leapfrog_step_sizes must be a sequence of length {K}, so that each proposal has its own "
                f"specified leapfrog step size, but instead found length of {len(sizes)}")
        for idx, step_size in enumerate(sizes):
            if not isinstance(step_size, float):
                raise TypeError(
                    f"each step size in leapfrog_step_sizes must be of type float, but found step size of "
                    f"type {type(step_size)} at index {idx}")
            if not step_size > 0:
                raise ValueError(
                    f"each step size in leapfrog_step_sizes must be positive, but found step size of "
                    f"{step_size} at index {idx}")
        counts = self._leapfrog_step_counts
        if not isinstance(counts, Sequence):
            raise TypeError(
                f"leapfrog_step_counts must be an instance of type sequence, but found type {type(counts)}")
        if len(counts) != K:
            raise ValueError(
                f"leapfrog_step_counts must be a sequence of length {K}, so that each proposal has its own "
                f"specified number of leapfrog steps, but instead found length of {len(counts)}")
        for idx, step_count in enumerate(counts):
            if not isinstance(step_count, int):
                raise TypeError(
                    f"each step count in leapfrog_step_counts must be of type int, but found step count of "
                    f"type {type(step_count)} at index {idx}")
            if not step_count > 0:
                raise ValueError(
                    f"each step count in leapfrog_step_counts must be positive, but found step count of "
                    f"{step_count} at index {idx}")
        damping = self._damping
        if not isinstance(damping, float):
            raise TypeError(f"damping must be of type float, but found type {type(damping)}")
        if not 0 < damping <= 1:
            raise ValueError(f"damping must be within (0, 1], but found damping of {damping}")

    def _state_tensors(self):
        return {"theta": self._theta_dc, "rho": self._rho_dc, "grad": self._grad, "lp": self._lp}

    def state_dict(self):
        if getattr(self, "_regrad", False) and self._have_cache:
            # the one-launch path keeps no gradient cache: bring self._grad up to date for the checkpoint (a sampler on
            # another path may load it); the values are the ones that path would have cached
            self._materialize(self._eval_grad(self._theta_dc, self._grad, None), self._grad)
            self._grad_calls -= 1  # (bookkeeping of the checkpoint, not a model call of the sampler's)
        return super().state_dict()

    def _state_extra(self):
        return {"rho_sign": self._rho_sign}

    def _load_extra(self, extra):
        self._rho_sign = float(extra.get("rho_sign", 1.0))
        self._draws_dev.fill_(int(self._draws))

    # -- views ----------------------------------------------------------------------------------------
    @property
    def _rho(self):
        r = self._rho_dc * self._rho_sign
        return r.t() if self._batched else np.array(r[:, 0].cpu().numpy())

    # -- lane-set helpers --------------------------------------------------------------------------------
    def _compact(self, mask, n, lvl):
        """Indices of the nonzero entries of mask[:n] in level lvl's idx buffer.
        Returns (count, idx or None when every lane is set)."""
        L = self._levels[lvl]
        self._ops.compact_indices(mask, n, L.idx, L.count)
        m = int(L.count.item())  # host needs the lane count to size the next launches
        return m, (None if m == n else L.idx)

    def _proposal_map(self, src, idx, n, k, lvl, tag):
        """Leapfrog (eps_k, L_k) + momentum flip from lanes idx of `src` into level lvl
        (drghmc.py:319-346, 253-289)."""
        ops, m = self._ops, self._metric_dev
        h, steps = float(self._leapfrog_step_sizes[k]), int(self._leapfrog_step_counts[k])
        dst = self._levels[lvl]
        th, rho, gbuf = dst.theta[:, :n], dst.rho[:, :n], dst.grad[:, :n]
        self._h_evals += steps
        self._h_lane_steps += steps * n
        self._h_stages.append((tag, n))
        if self._fused and self._model.bk_dr_proposal(src.theta, src.rho, src.grad, idx, th, rho, gbuf,
                                                      dst.logp[:n], dst.kin[:n], m, h, steps):
            return
        if self._traj_hook and self._model.bk_leapfrog_trajectory(src.theta, src.rho, src.grad, idx, th, rho, gbuf, dst.logp[:n],
                                                                  m, h, steps):
            self._grad_calls += steps
            ops.leapfrog_finish(rho, rho, gbuf, m, 0.5 * h, True, dst.kin[:n])
            return
        ops.first_step_gather(src.theta, src.rho, src.grad, idx, th, rho, m, h, 0.5 * h)
        for _ in range(steps - 1):
            if self._step_hook:
                self._grad_calls += 1
                self._model.bk_leapfrog_step(th, rho, m, h)
                continue
            g = self._eval_grad(th, gbuf, None)
            ops.kick_drift(th, th, rho, rho, g, m, h, False, 0.0, True, h)
        g = self._materialize(self._eval_grad(th, gbuf, dst.logp[:n]), gbuf)
        ops.leapfrog_finish(rho, rho, g, m, 0.5 * h, True, dst.kin[:n])

    def _accept(self, lvl, n, k, cur_h, cur_H, cur_idx, tag):
        """log acceptance probability of the level-lvl lanes against their parents
        (drghmc.py:391-446).  Returns level.a (valid for the first n lanes)."""
        ops = self._ops
        P = self._levels[lvl]
        ops.dr_level_begin(P.logp, P.kin, P.H, P.h, P.live, n)
        for i in range(k):
            if i == 0:
                m, sub = n, None  # dr_level_begin just set every lane live: no compaction, no host read
            else:
                m, sub = self._compact(P.live, n, lvl + 1)
            if m == 0:
                break
            gtag = "G%d(%s)" % (i, tag)
            self._proposal_map(_View(P, n), sub, m, i, lvl + 1, gtag)
            ga = self._accept(lvl + 1, m, i, P.h, P.H, sub, gtag)
            ops.dr_ghost_update(ga, sub, m, P.h, P.live, P.a)
        ops.dr_accept_prob(P.H, cur_H, P.h, cur_h, cur_idx, 1.0 if self._prob_retry else 0.0, P.live, P.a, n)
        return P.a

    # -- diagnostics fed by the draw itself ------------------------------------------------------------------
    def attach(self, moments=None, recorder=None):
        """Feed a ``RunningMoments`` and / or a ``DrawRecorder`` from inside every draw: their updates become part
        of the draw's launch sequence -- with device-side lane counts, of the ONE hipGraph a draw replays as -- and
        take their draw index from the sampler's device-side draw counter.  The objects are then up to date after
        every ``sample()`` / ``advance()`` without calls of their own (do not also call update() / record())."""
        for obj, kind in ((moments, "moments"), (recorder, "recorder")):
            if obj is not None:
                self._attached.append((kind, obj, self._draws - obj.n))
        self._drop_graphs()

    def detach(self):
        """Stop feeding the attached diagnostics (they keep what they have)."""
        self._attached = []
        self._drop_graphs()

    def _feed_attached(self, theta_dc, logp, on_device, defer=False):
        """Inside the draw: the attached diagnostics see the new state (theta_dc [D, C], joint log density).
        defer (device path, a draw that is followed by another one inside the same hipGraph): the first attached moments
        object and the first attached recorder are not fed here -- their calls are returned as ONE job for the next draw's
        generator launch (ops.diag_job: it reads theta_dc, the joint log density and the draw count as that draw finds them,
        i.e. as this draw leaves them)."""
        welford = record = None
        for kind, obj, off in self._attached:
            if on_device:
                if kind == "moments":
                    if defer and welford is None:
                        welford = obj._update_job(off)
                    else:
                        obj._update_dev(theta_dc, self._draws_dev, off)
                elif defer and record is None:
                    record = obj._record_job(theta_dc.shape[0], logp, off + 1)
                else:
                    obj._record_dev(theta_dc, logp, self._draws_dev, off + 1)
            elif kind == "moments":
                obj.update(theta_dc, layout="dc")
            else:
                obj.record(theta_dc, logp)
        if welford is None and record is None:
            return None
        return self._ops.diag_job(theta_dc, self._draws_dev, welford=welford, record=record)

    def _count_attached(self):
        for i, (kind, obj, off) in enumerate(self._attached):
            if kind == "recorder" and obj.n >= obj.series.shape[1]:
                raise IndexError("DrawRecorder is full")
            # the device-side update count / row index is (sampler's draw counter - off): keep it equal to the object's
            # own count whatever was restored since (load_state_dict of the sampler, of the object, in either order)
            now = self._draws - obj.n
            if now != off:
                self._attached[i] = (kind, obj, now)
                self._drop_graphs()  # (the offset is a scalar argument of the captured launches)

    DRAWS_PER_GRAPH = 10  # advance(n): draws replayed per hipGraph launch (each launch costs ~8 us between graphs)
    DEFER_MOMENTS = True  # ... and inside such a graph the attached moments / series of a draw ride on the next draw's generator launch

    def advance(self, n: int = 1):
        """n draws of every chain WITHOUT handing the state back (no copies): for runs whose draws are consumed
        by attached diagnostics only.  ``sample()`` = ``advance()`` + the returned (theta, logp) copies.
        Where a draw replays as a hipGraph, n > 1 replays graphs of up to DRAWS_PER_GRAPH consecutive draws: the same
        launches in the same order (a replay IS the next draws: every per-draw quantity lives on the device), without
        the ~8 us the device idles between two graph launches."""
        n = int(n)
        if n < 1:
            raise ValueError("advance(n): n >= 1")
        while n > 0:
            m = min(n, int(self.DRAWS_PER_GRAPH))
            # a multi-draw graph needs the settled regime: first evaluation done, momentum sign lazy, single draw captured
            if m > 1 and self._dev_counts and self._use_graph and self._graph is not None and self._rho_sign == -1.0 \
                    and self._have_cache and self._replay_many(m):
                n -= m
            else:
                self._step()
                n -= 1

    def _replay_many(self, m):
        """m draws as ONE graph replay; False (nothing done) where the single-draw path has work to do first."""
        self._count_attached()
        for kind, obj, off in self._attached:
            if kind == "recorder" and obj.n + m > obj.series.shape[1]:
                return False
        if (self._graph is None or self._graph_key() != self._graph_scalars
                or tuple(int(k) for k in self._leapfrog_step_counts) != self._schedule_counts):
            return False  # (_step() captures again / rebuilds the schedule)
        g = self._graph_many.get(m)
        if g is None:
            import gc

            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            gc_was_on = gc.isenabled()
            gc.disable()  # (see _run_draw: no collection while the stream is capturing)
            try:
                with torch.cuda.graph(g):
                    # the moments update of every draw but the last rides on the NEXT draw's generator launch (memory-bound
                    # beside issue-bound: the longer of the two times instead of their sum); the last one is its own launch,
                    # so the attached objects are up to date when advance() returns
                    job = None
                    for i in range(m):
                        job = self._draw_dev(pending=job, defer=self.DEFER_MOMENTS and i + 1 < m)
            finally:
                if gc_was_on:
                    gc.enable()
            self._graph_many[m] = g
        g.replay()
        self._first_eval = 0
        self._draws += m
        for kind, obj, off in self._attached:
            obj.n += m
        return True

    def _drop_graphs(self):
        super()._drop_graphs()
        self._graph_many = {}

    def _graph_key(self):
        return (float(self._damping), float(self._rho_sign), bool(self._prob_retry),
                tuple(float(h) for h in self._leapfrog_step_sizes), tuple(int(n) for n in self._leapfrog_step_counts))

    # -- one draw for every chain -----------------------------------------------------------------------------
    def sample(self):
        self._step()
        return self._draw_out(self._theta_dc, self._cur_H)

    def _step(self):
        self._count_attached()
        if self._dev_counts:
            if not self._have_cache:  # drghmc.py:243-245 (first draw only; outside any capture)
                self._materialize(self._eval_grad(self._theta_dc, self._grad, self._lp), self._grad)
                self._have_cache = True
                self._first_eval = 1
            else:
                self._first_eval = 0
            if tuple(int(n) for n in self._leapfrog_step_counts) != self._schedule_counts:
                # the trajectories change length: close the books of the old schedule (one host read, rare)
                self._steps_total_base = float(self.lane_steps_total.item())
                self._make_schedule()
                self._drop_graphs()
            self._run_draw(self._draw_dev)
            self._rho_sign = -1.0  # drghmc.py:388, applied lazily
            self._draws += 1
            for kind, obj, off in self._attached:
                obj.n += 1
            return
        self._step_host()
        self._feed_attached(self._theta_dc, self._cur_H, False)

    def _step_host(self):
        ops = self._ops
        C, m = self._C, self._metric_dev
        self._h_evals, self._h_lane_steps, self._h_stages = 0, 0, []
        damping = self._damping
        # partial momentum refresh (drghmc.py:360-364); the stored momentum may still carry the
        # previous draw's pending flip (drghmc.py:388), applied here through the sign of loc_mul
        ops.momentum_refresh(self._rng_kind, self._rng_state, self._rho_dc,
                             self._rho_sign * math.sqrt(1 - damping), math.sqrt(damping), self._rho_dc, m,
                             self._kin, None, self._rng_work)
        self._rho_sign = 1.0
        if not self._have_cache:  # drghmc.py:243-245 (first draw only)
            self._materialize(self._eval_grad(self._theta_dc, self._grad, self._lp), self._grad)
            self._have_cache = True
            self._h_evals += 1
        ops.dr_begin(self._lp, self._kin, self._cur_H, self._cur_h, self._rej, self._alive)
        cur = _Cur(self)
        pr = 1.0 if self._prob_retry else 0.0
        P0 = self._levels[0]
        for k in range(int(self._max_proposals)):
            ops.dr_retry_test(self._rng_kind, self._rng_state, self._rej, pr, self._alive)  # :369-371
            if k == 0:
                # reject_logp = 0 at the first stage, so log(u) < 0 always holds (drghmc.py:366-371):
                # every chain proposes; the uniform is still drawn above (stream alignment)
                n, idx = C, None
            else:
                n, idx = self._compact(self._alive, C, 0)
            if n == 0:
                break
            tag = "P%d" % k
            self._proposal_map(cur, idx, n, k, 0, tag)                                    # :373
            a = self._accept(0, n, k, self._cur_h, self._cur_H, idx, tag)                 # :374-376
            ops.dr_accept_test(self._rng_kind, self._rng_state, idx, a, P0.H, n, self._cur_H, self._cur_h,
                               self._rej, self._alive, P0.accepted)                        # :378-385
            ops.scatter_columns(P0.accepted, idx, n,
                                [self._theta_dc, self._rho_dc, self._grad],
                                [P0.theta[:, :n], P0.rho[:, :n], P0.grad[:, :n]], self._lp, P0.logp)
        self._rho_sign = -1.0  # drghmc.py:388, applied lazily
        self._draws += 1
        self._draws_dev.fill_(self._draws)
        self._host_stats = (self._h_evals, self._h_lane_steps, self._h_stages)

    # -- the same draw with lane counts on the device: a fixed launch sequence ----------------------------
    def _proposal_dev(self, src, idx, n_dev, k, lvl, job=None, ghost=None, own_ghosts=0):
        """Proposal k from lanes idx (count n_dev; None = all C chains) of `src` into level lvl; `job`: a
        scatter job the launch carries along; `ghost`: the level's accept probability + parent update, done by
        the same launch (a ghost without ghosts of its own).  own_ghosts >= 1: the produced level has ghosts of
        its own, and the launch runs the first of them as well (every produced lane has one, it starts from the
        registers the proposal ends in; bk_ghost0).  Returns then (True, list of the lanes that go on to ghost 1
        or None), else (False, None)."""
        h, steps = float(self._leapfrog_step_sizes[k]), int(self._leapfrog_step_counts[k])
        dst = self._levels[lvl]
        slot = self._slot
        self._slot += 1
        if not self._one_launch:
            return self._proposal_steps_dev(src, idx, n_dev, h, steps, dst, slot, ghost)
        g0, fused, following = None, False, None
        if own_ghosts >= 1 and self._fuse_first_ghost:
            fused = True
            gslot = self._slot  # (the schedule lists G0(X) right after X)
            self._slot += 1
            if own_ghosts >= 2:
                following = (self._levels[lvl + 1].idx, self._new_list())
            g0 = self._ops.ghost0(self._leapfrog_step_sizes[0], self._leapfrog_step_counts[0], dst.a,
                                  1.0 if self._prob_retry else 0.0, *(following or (None, None)),
                                  lanes_out=self._slot_lanes[gslot:gslot + 1],
                                  lanes_total=self._slot_lanes_total[gslot:gslot + 1])
        # the launch also counts its lanes (per draw and in total) and sets up its level for accept()
        kw = {} if g0 is None else {"ghost0": g0}
        if ghost is not None:
            par, pr, nxt_list = ghost
            ghost = self._ops.ghost_link(par.H, par.h, par.live, par.a, dst.a, pr, *(nxt_list or (None, None)))
        ok = self._model.bk_dr_proposal(src.theta, src.rho, None if self._regrad else src.grad, idx, dst.theta, dst.rho,
                                        None if self._regrad else dst.grad, dst.logp,
                                        dst.kin, self._metric_dev, h, steps, n_dev=n_dev,
                                        lanes_out=self._slot_lanes[slot:slot + 1],
                                        lanes_total=self._slot_lanes_total[slot:slot + 1],
                                        level=(dst.H, dst.h, dst.live), job=job, ghost=ghost, **kw)
        assert ok
        return fused, following

    def _proposal_steps_dev(self, src, idx, n_dev, h, steps, dst, slot, ghost):
        """The same proposal as the reference's own sequence of model calls (drghmc.py:273-288), every launch
        taking the lane count from the device: gathering first step, (steps - 1) x {counted gradient op,
        kick + drift}, last gradient + log density, final half-kick + flip + kinetic energy -- whose launch also
        sets the level up for accept() and counts the lanes -- and, for a ghost without ghosts of its own, its
        acceptance probability + parent update.  2 * steps + 1 (+ 1) launches, no host read."""
        ops, m, model, C = self._ops, self._metric_dev, self._model, self._C
        th, rho, g = dst.theta, dst.rho, dst.grad
        self._grad_calls += steps
        if not (self._traj_hook and model.bk_leapfrog_trajectory(src.theta, src.rho, src.grad, idx, th, rho, g, dst.logp, m, h,
                                                                 steps, n_dev=n_dev)):                  # :276-285, one launch
            ops.first_step_gather(src.theta, src.rho, src.grad, idx, th, rho, m, h, 0.5 * h, n_dev=n_dev)   # :276-278
            for _ in range(steps - 1):                                                                      # :280-283
                if self._step_hook:
                    model.bk_leapfrog_step(th, rho, m, h, n_dev)
                    continue
                model.bk_eval(th, g, None, n_dev)
                ops.kick_drift(th, th, rho, rho, g, m, h, False, 0.0, True, h, n_dev=n_dev)
            model.bk_eval(th, g, dst.logp, n_dev)                                                           # :285
        ops.leapfrog_finish(rho, rho, g, m, 0.5 * h, True, dst.kin, n_dev=n_dev,                        # :286, :345
                            level=(dst.logp, dst.H, dst.h, dst.live), lanes_out=self._slot_lanes[slot:slot + 1],
                            lanes_total=self._slot_lanes_total[slot:slot + 1])
        if ghost is not None:
            par, pr, nxt_list = ghost
            args = (dst.H, par.H, dst.h, par.h, idx, pr, dst.live, dst.a, C, par.live, par.a)          # :426-446
            if nxt_list is None:
                ops.dr_accept_prob_ghost(*args, n_dev=n_dev)
            else:
                ops.dr_accept_prob_ghost_next(*args, nxt_list[0], nxt_list[1], n_dev=n_dev)
        return False, None

    def _new_list(self):
        """The next unused lane counter of this draw (zeroed by bk_dr_begin_retry)."""
        i = self._list
        self._list += 1
        if i >= self._counters.numel():
            raise RuntimeError("max_proposals too large for the device-side lane lists")
        return self._counters[i:i + 1]

    def _accept_dev(self, lvl, n_dev, k, parent=None, sub=None, parent_next=None, first=(False, None)):
        """_accept() over lane sets whose sizes live on the device (n_dev None = all C chains): the ghost
        proposals of level `lvl` and their recursive accepts.  For a ghost level (`parent` given: the level
        its lanes belong to, paired by `sub`) the level's own accept probability and the update of the
        parent are one launch (bk_dr_accept_prob_ghost); for the stage's proposal itself (level 0) the
        probability is evaluated by the accept test's launch (bk_dr_accept_prob_test, in _draw_dev).
        The lane set of ghost i >= 1 -- the lanes of this level that are still live -- is listed by the launch
        that updates the level after ghost i - 1 (`parent_next` = (list, counter) handed down to it).
        `first`: what _proposal_dev returned for this level -- ghost 0 already run by the level's own launch."""
        ops, C = self._ops, self._C
        P = self._levels[lvl]  # (H, h, live of the level were set by the proposal's own launch)
        nxt = self._levels[lvl + 1] if lvl + 1 < len(self._levels) else None
        m_dev, gsub = n_dev, None  # ghost 0: every lane of the level is still live
        for i in range(k):
            following = None
            if i == 0 and first[0]:
                if first[1] is not None:
                    gsub, m_dev = first[1]
                continue
            if i + 1 < k:  # the list ghost i + 1 will run over, built while ghost i's result is applied
                buf = nxt.idx if (i % 2 == 0) else nxt.idx_alt
                following = (buf, self._new_list())
            if i == 0 or (i == 1 and self._fuse_first_ghost):
                # a ghost of the first proposal kind has no ghosts of its own, one of the second kind has one and
                # its launch runs it: the ghost's acceptance probability and the update of this level (:426-446)
                # are done by the ghost's launch
                pr = 1.0 if self._prob_retry else 0.0
                self._proposal_dev(P, gsub, m_dev, i, lvl + 1, ghost=(P, pr, following), own_ghosts=i)
            else:
                done = self._proposal_dev(P, gsub, m_dev, i, lvl + 1, own_ghosts=i)
                self._accept_dev(lvl + 1, m_dev, i, parent=P, sub=gsub, parent_next=following, first=done)
            if following is not None:
                gsub, m_dev = following
        if parent is not None:
            pr = 1.0 if self._prob_retry else 0.0
            if parent_next is None:
                ops.dr_accept_prob_ghost(P.H, parent.H, P.h, parent.h, sub, pr, P.live, P.a, C, parent.live, parent.a,
                                         n_dev=n_dev)                                         # :426-446
            else:
                ops.dr_accept_prob_ghost_next(P.H, parent.H, P.h, parent.h, sub, pr, P.live, P.a, C, parent.live,
                                              parent.a, parent_next[0], parent_next[1], n_dev=n_dev)

    def _draw_dev(self, pending=None, defer=False):
        """One draw as a fixed launch sequence.  pending: the previous draw's deferred moments update (a welford job), done
        by workgroups of this draw's generator launch; defer: hand this draw's update to the next one the same way (returned)."""
        ops = self._ops
        C, m, damping = self._C, self._metric_dev, self._damping
        pr = 1.0 if self._prob_retry else 0.0
        # partial momentum refresh + its kinetic energy :360-364, start of the draw + the first stage's retry test
        # (always passed, its uniform drawn) :365-371: the generator's launch + one more
        ops.dr_refresh_begin(self._rng_kind, self._rng_state, self._rho_dc, self._rho_sign * math.sqrt(1 - damping),
                             math.sqrt(damping), self._rho_dc, m, self._kin, self._rng_work, self._lp, self._cur_H,
                             self._cur_h, self._rej, self._alive, pr, self._counters, self._draws_dev, side=pending)
        cur = _Cur(self)
        self._slot = 0
        self._list = 0
        K = int(self._max_proposals)
        n_dev, idx = None, None  # every chain proposes at the first stage
        # The stages alternate between two level-0 buffer sets, so that the scatter of stage k's accepted
        # proposals into the chains' current point can run INSIDE stage k+1's proposal launch (surplus
        # workgroups beside a sparse, latency-bound lane set): it reads stage k's buffers and writes columns
        # of accepted chains, the proposal reads columns of rejected chains and writes the other set.
        # (a step-by-step proposal carries no job: one buffer set, the scatter is a launch of its own)
        level0 = [self._levels[0], self._level0_alt if self._one_launch else self._levels[0]]
        job = None
        for k in range(K):
            P0 = self._levels[0] = level0[k % 2]
            nidx, ncount = (None, None) if k + 1 == K else ((P0.idx if k % 2 == 0 else P0.idx_alt), self._new_list())
            done = self._proposal_dev(cur, idx, n_dev, k, 0, job=job, own_ghosts=k)               # :373
            self._accept_dev(0, n_dev, k, first=done)                                             # :374-376
            # accept test :441-446, :378-385; for the rejected chains the next stage's retry test :369-371;
            # the chains that propose again are listed for the next stage
            if k + 1 < K:
                ops.dr_accept_prob_test_next(self._rng_kind, self._rng_state, idx, P0.H, P0.h, P0.live, P0.a, pr, C,
                                             self._cur_H, self._cur_h, self._rej, self._alive, P0.accepted, nidx,
                                             ncount, n_dev=n_dev)
            else:
                ops.dr_accept_prob_test(self._rng_kind, self._rng_state, idx, P0.H, P0.h, P0.live, P0.a, pr, C,
                                        self._cur_H, self._cur_h, self._rej, self._alive, P0.accepted, n_dev=n_dev)
            moved = ([self._theta_dc, self._rho_dc], [P0.theta, P0.rho]) if self._regrad else \
                ([self._theta_dc, self._rho_dc, self._grad], [P0.theta, P0.rho, P0.grad])
            scatter = (P0.accepted, idx, C, moved[0], moved[1], self._lp, P0.logp)               # :379-381
            if k + 1 < K and self._one_launch:
                job = ops.scatter_job(*scatter, n_dev=n_dev)   # rides on the next stage's proposal launch
            else:
                ops.scatter_columns(*scatter, n_dev=n_dev)
            if k + 1 < K:
                idx, n_dev = nidx, ncount
        self._levels[0] = level0[0]
        return self._feed_attached(self._theta_dc, self._cur_H, True, defer=defer)

    @property
    def lane_steps_total(self):
        """Chain-steps run since construction (0-d device tensor; the per-trajectory lane counts are
        accumulated by the proposal launches themselves)."""
        return (self._slot_lanes_total.to(torch.float64) * self._slot_steps).sum() + self._steps_total_base


class _View:
    """First n lanes of a level, as a source of phase-space points."""

    def __init__(self, level, n):
        self.theta, self.rho, self.grad = level.theta[:, :n], level.rho[:, :n], level.grad[:, :n]


class _Cur:
    """The chains' current points as a source of phase-space points."""

    def __init__(self, s):
        self.theta, self.rho, self.grad = s._theta_dc, s._rho_dc, s._grad
