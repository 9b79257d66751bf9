"""Many-chain Metropolis-adjusted Langevin on the GPU: drop-in for ``bayes_kit/mala.py:14-79``.

Per draw and chain: theta' = (theta + eps*grad) + sqrt(2 eps) z   [mala.py:41-45];
one gradient call at theta' [:46-48]; forward / reverse proposal densities [:50-53,68-79];
Metropolis-Hastings accept with strict ``<`` [metropolis.py:70-76]; (logp, grad) of the
current point are cached [mala.py:31-32,62-64]; the MODEL log density is returned [:66].

Two ways through the device:

* **two-pass** (many chains, Philox streams, 32 <= D <= 1024): a draw is the model's gradient op
  plus ONE kernel (``bk_mala_step``) that keeps a block of 16 chains x all dimensions on chip
  between the per-chain sums and the select, and also writes the NEXT draw's proposal from
  normals generated ahead on a side stream -- 88*D bytes of HBM traffic per chain-draw.  The
  state array is rebound every draw, as the reference rebinds ``_theta`` (mala.py:62): the
  tensor ``sample()`` returns IS the new state and is never written again.
  For a model that is a SEPARABLE density the library can inline (``bk_mala_step``: the built-in Gaussians, an elementwise
  ``CTarget.from_source`` / traced ``TorchModel``) the step kernel recomputes both gradients from theta and theta' and stores
  none: the model's launch is its log density alone and a draw moves 56*D bytes; the same draws bit for bit
  (``path="opaque"`` keeps the model-opaque pair of launches).
* **step by step** (everything else: a single-chain host model, PCG64 streams, odd shapes):
  proposal, gradient, proposal densities, accept and select as separate kernels.

Both consume each chain's stream in the reference's order (D normals, then one uniform) and give
the same draws bit for bit.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._engine import ManyChainSampler


class MALA(ManyChainSampler):
    # 65,536 chains: 512-KiB rows.  The one-pass step kernel keeps 16 chains x all dims on a CU (1,024 row pieces of 128
    # bytes, all at the same offset of a 4-KiB-aligned pitch) and the gradient + log density op walks down the rows too:
    # 1.39 -> 1.27 ms per draw with the rows 1,152 bytes further apart (1.57 -> 1.38 on a box whose allocations sit
    # worse; pads 48..400 columns within 2 % of each other, 0 / 8 / 32 / 528 are the bad ones).
    STATE_PAD_COLUMNS = 144
    SINGLE_LAUNCH_MAX_DIMS = 4096  # (one lane walks the coordinates: the single-chain mode is for small models)

    TUNING = ("graph", "prefetch_rng", "tune_placement", "two_pass", "single_launch")

    def __init__(self, model, epsilon: float, init=None, seed=None, *, chains: Optional[int] = None,
                 chain_id0: int = 0, path: str = "auto", tuning: Optional[dict] = None, ops=None, **knobs):
        """The reference's arguments (mala.py:15-21), then the engine's (ManyChainSampler: chains, chain_id0, path, tuning,
        ops; path "step" and "opaque" both keep the model-opaque pair {gradient op, step kernel}).  Tuning knobs (none
        changes a result): ``graph``, ``prefetch_rng``, ``tune_placement``, ``two_pass`` (gradient op + ONE step kernel
        per draw; default where the shape allows), ``single_launch`` (one chain: one launch per draw)."""
        fuse_builtin, _ = self._resolve_path(path)
        tn = self._resolve_tuning(tuning, knobs)
        graph, prefetch_rng, tune_placement = tn.get("graph"), tn.get("prefetch_rng"), tn.get("tune_placement")
        two_pass, single_launch = tn.get("two_pass"), tn.get("single_launch")
        self._epsilon = epsilon
        self._setup(model, None, init, seed, chains, chain_id0, ops)
        self._init_graph(graph, prefetch_rng)
        D, C, dev = self._dim, self._C, self._ops.device
        f64 = dict(dtype=torch.float64, device=dev)
        self._theta_p = self._new_state()
        self._grad = self._new_state()
        self._grad_p = self._new_state()
        self._lp = torch.empty(C, **f64)
        self._lp_p = torch.empty(C, **f64)
        self._fwd = torch.empty(C, **f64)
        self._rev = torch.empty(C, **f64)
        self._logu = torch.empty(C, **f64)
        self._ret = torch.empty(C, **f64)
        self._mask = torch.empty(C, dtype=torch.uint8, device=dev)
        self._accepted = torch.zeros(1, dtype=torch.int32, device=dev)
        self._draws = 0
        # As in HMCDiag: the D proposal normals and the accept uniform of draw n+1 do not depend
        # on draw n and are consumed in a fixed order (mala.py:44 then metropolis.py:74), so they
        # are generated on a second HIP stream under draw n's HBM-bound kernels.
        # With graph=True the randomness is generated in line (a captured draw stays a linear graph,
        # see ManyChainSampler), whatever prefetch_rng says; the draws are bit-identical either way.
        if prefetch_rng is None:
            prefetch_rng = self._batched and dev.type == "cuda"
        self._prefetch = bool(prefetch_rng) and self._batched and not self._use_graph
        self._pf_slot, self._pf_ready, self._pf_event = 0, False, None
        # Philox streams: the normals come from the wavefront-per-chain generator, chain-major
        # (zt[c, d]); the consuming kernel turns them through LDS.  Otherwise (PCG64 host-seeded
        # single chains, tiny D): one lane per chain, normals in the state layout.
        self._chain_major = self._rng_kind == _lib.RNG_PHILOX and D >= 32
        ld = self._theta_dc.stride(0) if D > 1 else C
        can_two_pass = self._batched and self._chain_major and self._ops.mala_step_supported(C, D, ld)
        if two_pass and not can_two_pass:
            raise ValueError("two_pass=True needs a batched model, Philox streams, 32 <= D <= 1024 and an even number "
                             "of chains")
        self._two_pass = can_two_pass if two_pass is None else bool(two_pass)
        self.path = "two-pass (bk_mala_step)" if self._two_pass else "step-by-step"
        # a separable density the library can inline: the step kernel recomputes the gradients (none is stored between draws;
        # _grad is brought up to date when somebody asks for it)
        self._sep_step = bool(fuse_builtin) and self._two_pass and hasattr(model, "bk_mala_step") and hasattr(model, "bk_eval")
        self._grad_stale = False
        if self._sep_step:
            self.path = "two-pass, gradients recomputed in the step kernel (model.bk_mala_step)"
        nbuf = 2 if self._prefetch else 1
        if self._chain_major:
            dp = (D + 7) // 8 * 8
            self._zt_bufs = [torch.empty((C, dp), **f64) for _ in range(nbuf)]
            self._z_bufs = [zt[:, :D].t() for zt in self._zt_bufs]
        elif self._prefetch:
            self._z_bufs = [self._new_state() for _ in range(2)]
        if self._prefetch or self._two_pass:
            self._logu_bufs = [torch.empty(C, **f64) for _ in range(nbuf)]
        if self._prefetch:
            self._init_side_stream()
            self._rng_logical = self._rng_state.clone()
        # two-pass pipeline: theta_p holds the proposal of the NEXT draw (made by the previous
        # draw's kernel from normals generated one draw ahead).  A generation unit is
        # (U_n, Z_{n+1}): the accept uniform of draw n, then the D normals of draw n+1 -- stream
        # order.  The LOGICAL stream position after draw n lies inside unit n (after U_n): it is
        # snapshotted there, per double-buffer slot.
        self._pipe_valid = False   # theta_p is the proposal of the next draw
        self._unit_ready = False   # slot _pf_slot holds the next draw's unit (prefetch)
        self._cur_slot = 0         # slot whose unit the last draw consumed
        if self._two_pass:
            self._snap = [torch.empty_like(self._rng_state) for _ in range(nbuf)]
        # ONE chain driven by a reference-style host model (the README example, config 1): after the model call a draw is
        # ONE launch (bk_mala_single_draw: proposal densities, accept test, select AND the next draw's proposal), the
        # model's outputs go in and the draw comes out through pinned host memory the device addresses directly, and
        # the host waits on a sequence word instead of synchronising the stream: 1 launch + 1 model call per draw
        # (the step-by-step path: 6 launches, 2 device-to-host and 1 host-to-device copies, 121 us per draw).
        can_single = (not self._batched) and dev.type == "cuda" and D <= self.SINGLE_LAUNCH_MAX_DIMS
        if single_launch and not can_single:
            raise ValueError("single_launch=True needs a single-chain host model on a GPU and at most "
                             f"{self.SINGLE_LAUNCH_MAX_DIMS} dimensions")
        self._single = can_single if single_launch is None else bool(single_launch)
        if self._single:
            self.path = "single chain, one launch per draw (bk_mala_single_draw)"
            self._h_in = torch.zeros(D + 1, dtype=torch.float64).pin_memory()
            self._h_out = torch.zeros(2 * D + 3, dtype=torch.float64).pin_memory()
            self._h_in_np, self._h_out_np = self._h_in.numpy(), self._h_out.numpy()
            self._snap1 = self._rng_state.clone()  # (a generator's store() leaves the key words alone)
            self._seq = 0.0
            self._single_fn = self._ops.lib.bk_mala_single_draw
        self.placement = None
        if self._wants_placement_tuning(tune_placement) and not self._two_pass:
            self._tune_placement()
        # mala.py:31-32: (logp, grad) at theta0
        self._materialize(self._eval_grad(self._theta_dc, self._grad, self._lp), self._grad)

    def _fresh_grad(self):
        """The cached gradient of the current point; the separable step kernel does not keep it (recomputed on demand)."""
        if self._grad_stale:
            self._materialize(self._eval_grad(self._theta_dc, self._grad, None), self._grad)
            self._grad_stale = False
        return self._grad

    def refresh_cache(self):
        """Recompute the cached (logp, grad) of the current point (mala.py:31-32) after ``_theta`` was
        assigned or edited from outside, as the reference's constructor does for ``init``; a proposal
        made ahead from the old point is discarded (its normals are drawn again, same values)."""
        self._invalidate_pipe(restore_stream=True)
        self._materialize(self._eval_grad(self._theta_dc, self._grad, self._lp), self._grad)
        self._grad_stale = False

    def _invalidate_pipe(self, restore_stream):
        if (self._two_pass or getattr(self, "_single", False)) and self._pipe_valid and restore_stream:
            self._rng_state.copy_(self._logical_rng())  # un-consume what was generated ahead
        self._pipe_valid, self._unit_ready, self._pf_event = False, False, None
        self._drop_graphs()

    def _tune_placement(self):
        """Roles (theta', grad, grad') for the proposal, proposal-density and select kernels, which
        stream four to five arrays at equal offsets: see ManyChainSampler._tune_roles."""
        ops, th, eps = self._ops, self._theta_dc, float(self._epsilon)
        z = self._z_bufs[0] if hasattr(self, "_z_bufs") else None
        out = self._new_state()
        self._mask.fill_(1)

        def cost(a):
            thp, g, gp = a
            ms = self._time_ms(lambda: ops.mala_logq(th, g, thp, gp, eps, self._fwd, self._rev))
            ms += self._time_ms(lambda: ops.select_columns(self._mask, th, thp, g, gp, out))
            if z is not None:
                ms += self._time_ms(lambda: ops.mala_propose_from_normals(th, g, z, thp, eps, math.sqrt(2 * eps)))
            return ms

        keep = th.clone()  # the select kernel writes theta: restore the initial state afterwards
        (self._theta_p, self._grad, self._grad_p), rep = self._tune_roles([self._theta_p, self._grad, self._grad_p],
                                                                          cost)
        th.copy_(keep)
        self.placement = {"draw_kernels_ms_as_allocated": rep["ms_as_allocated"],
                          "draw_kernels_ms_chosen": rep["ms_chosen"], "assignments_tried": rep["assignments_tried"]}

    def _state_tensors(self):
        return {"theta": self._theta_dc, "grad": self._fresh_grad(), "lp": self._lp, "accepted": self._accepted}

    def _logical_rng(self):
        if getattr(self, "_single", False):
            if not self._pipe_valid:
                return self._rng_state
            torch.cuda.synchronize()
            return self._snap1  # (the next draw's normals are already consumed: the position after this draw's uniform)
        if self._two_pass:
            if not self._pipe_valid:
                return self._rng_state
            if self._ops.device.type == "cuda":
                torch.cuda.synchronize()
            return self._snap[self._cur_slot]
        if self._prefetch and self._pf_ready:
            if self._pf_event is not None:
                self._pf_event.synchronize()
            return self._rng_logical
        return self._rng_state

    def rng_state(self):
        return self._logical_rng().cpu().numpy().view(np.uint64)

    def load_state_dict(self, sd):
        if self._two_pass:
            # draws handed out earlier alias past state arrays: restore into a fresh one
            self._theta_dc = self._new_state()
        super().load_state_dict(sd)

    def _after_load(self):
        self._grad_stale = False  # (the gradient of the restored point came with it)
        self._pf_event, self._pf_slot, self._pf_ready = None, 0, False
        self._invalidate_pipe(restore_stream=False)  # the stream was just restored to its logical position

    def _gen(self, slot):
        if self._chain_major:
            self._ops.normals_chain_major(self._rng_kind, self._rng_state, self._zt_bufs[slot], self._dim)
        else:
            self._ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, self._z_bufs[slot],
                                       None, None)
        self._ops.log_uniform(self._rng_kind, self._rng_state, self._logu_bufs[slot])

    def _take_randomness(self):
        main = torch.cuda.current_stream()
        cur, nxt = self._pf_slot, 1 - self._pf_slot
        if not self._pf_ready:
            self._gen(cur)
        elif self._pf_event is not None:
            main.wait_event(self._pf_event)
        ready, ev = self._ev_ready[nxt], self._ev_done[nxt]
        ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._rng_logical.copy_(self._rng_state)
            self._gen(nxt)
            ev.record(self._side)
        self._pf_event, self._pf_slot, self._pf_ready = ev, nxt, True
        return self._z_bufs[cur], self._logu_bufs[cur]

    # -- two-pass: generation units (U_n, Z_{n+1}) ------------------------------------------------------
    def _gen_unit(self, slot):
        ops = self._ops
        ops.log_uniform(self._rng_kind, self._rng_state, self._logu_bufs[slot])     # U_n   [metropolis.py:74]
        # Z_{n+1} [mala.py:44]; the generator leaves the table as it found it -- the position after draw n
        # -- in snap[slot]
        ops.normals_chain_major(self._rng_kind, self._rng_state, self._zt_bufs[slot], self._dim, self._snap[slot],
                                max_workgroups=self.generator_workgroups if self._prefetch else 0)

    def _take_unit(self):
        """(log u of this draw, chain-major normals of the next); with prefetch also starts the
        unit after that on the side stream."""
        if not self._prefetch:
            self._gen_unit(0)
            self._cur_slot = 0
            return self._logu_bufs[0], self._zt_bufs[0]
        main = torch.cuda.current_stream()
        cur, nxt = self._pf_slot, 1 - self._pf_slot
        if not self._unit_ready:
            self._gen_unit(cur)  # first draw after a (re)start: nothing generated ahead yet
        elif self._pf_event is not None:
            main.wait_event(self._pf_event)
        self._pf_slot, self._unit_ready, self._cur_slot = nxt, True, cur
        if self.generate_with == "grad":
            self._start_unit(nxt)
        return self._logu_bufs[cur], self._zt_bufs[cur]

    def _start_unit(self, nxt, recorded=False):
        """Queue the generator of the next unit on the side stream, behind everything queued on the main
        stream so far (or up to where the caller recorded the slot's ready event).  Slot nxt was last read by
        the previous draw's kernel, already queued on `main`; the RNG table is shared, so the side stream also
        starts after anything generated in line."""
        main = torch.cuda.current_stream()
        ready, ev = self._ev_ready[nxt], self._ev_done[nxt]
        if not recorded:
            ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._gen_unit(nxt)
            ev.record(self._side)
        self._pf_event = ev

    # Whether the step kernel waits for the generator of the next unit (queued on the side stream with the model's
    # launch).  Rounds 2-5: yes -- the generator needed 169 registers per lane, could not sit beside a step workgroup
    # (192 x 512 threads per CU), and the two only slowed each other down (1.52 against 1.3 ms per draw).  Round 6's
    # generator needs 93 and finishes in 0.26-0.29 ms: letting its tail run under the step kernel is the fastest of the
    # 32 schedules measured (65,536 x 1024, one box: model-opaque 1.124 ms against 1.152 serialized, density inlined
    # 0.802 against 0.876; profiles/r6_mala.md).  Results do not depend on it (events order the data).
    serialize_step = False
    # When the generator of the next unit starts: "grad" = with the model's gradient op (then, with
    # serialize_step, the step kernel waits for it), "step" = together with the step kernel, behind the
    # gradient op (the step kernel's workgroups are queued first and take one slot per CU; a generator
    # workgroup fits beside each: 160 of the 256 free registers per SIMD lane, 6 of the 27 free KB of LDS).
    generate_with = "grad"
    # Workgroups of the side-stream generator (0 = one per 16 chains, the kernel's own grid).  A bound of one per CU
    # turns it into a background kernel with one wavefront per SIMD, beside which a step workgroup still fits.
    generator_workgroups = 0
    step_first = True  # (experiments: with generate_with = "step", which of the two is queued first)

    def accept_rate(self) -> float:
        n = self._draws * self._C
        return float(self._accepted.item()) / n if n else float("nan")

    @property
    def last_accept(self):
        return self._mask.bool() if self._batched else bool(self._mask[0].item())

    @property
    def _log_p_theta(self):
        return self._lp if self._batched else float(self._lp[0].item())

    @property
    def _log_p_grad_theta(self):
        return self._fresh_grad().t() if self._batched else self._grad[:, 0].cpu().numpy()

    def _graph_key(self):
        return float(self._epsilon)

    # -- one chain, one launch per draw ---------------------------------------------------------------------------
    def _launch_single(self, have_prop):
        self._seq += 1.0
        eps = float(self._epsilon)
        rc = self._single_fn(self._rng_kind, self._rng_state.data_ptr(), self._rng_state.stride(0),
                             self._theta_dc.data_ptr(), self._grad.data_ptr(), self._lp.data_ptr(), self._theta_p.data_ptr(),
                             self._h_in.data_ptr(), self._h_out.data_ptr(), self._snap1.data_ptr(), self._snap1.stride(0),
                             self._mask.data_ptr(), self._accepted.data_ptr(), eps, math.sqrt(2 * eps), self._dim,
                             int(have_prop), self._seq, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "bk_mala_single_draw")
        # wait for the launch's last store (the sequence word, released at system scope behind everything else)
        flag, i, want = self._h_out_np, 2 * self._dim + 2, self._seq
        for _ in range(200000):
            if flag[i] == want:
                return
        torch.cuda.current_stream().synchronize()  # (a slow box, or an error that the synchronisation reports)
        if flag[i] != want:
            raise _lib.BkHipError("bk_mala_single_draw did not complete")

    def _sample_single(self):
        D = self._dim
        if not self._pipe_valid:
            self._launch_single(0)  # this draw's proposal from the stream's next D normals (mala.py:41-45)
            self._pipe_valid = True
        hin, hout = self._h_in_np, self._h_out_np
        self._grad_calls += 1
        lp, g = self._model.log_density_gradient(np.array(hout[D + 2:2 * D + 2]))             # mala.py:46-48
        hin[1:] = np.broadcast_to(np.asarray(g, dtype=np.float64), (D,))  # array-likes allowed (mala.py:32)
        hin[0] = float(lp)
        self._launch_single(1)                                                                 # mala.py:50-66, then :41-45
        self._draws += 1
        return np.array(hout[:D]), np.float64(hout[D])

    def sample(self):
        if self._single:
            return self._sample_single()
        self._run_draw(self._draw2 if self._two_pass else self._draw)
        if self._sep_step:
            self._grad_stale = True  # (here, not in _draw2: a replayed hipGraph does not run the Python of the draw)
        self._join_side_stream()
        self._draws += 1
        if self._two_pass and not self._use_graph:
            return self._theta_dc.t(), self._ret.clone()  # the new state array itself (never written again)
        return self._draw_out(self._theta_dc, self._ret)

    def _draw2(self):
        ops = self._ops
        eps = float(self._epsilon)
        s2 = math.sqrt(2 * eps)
        th, thp = self._theta_dc, self._theta_p
        if not self._pipe_valid:
            # this draw's proposal from the stream's next D normals (mala.py:41-45); later draws'
            # proposals are written by the previous draw's kernel
            ops.normals_chain_major(self._rng_kind, self._rng_state, self._zt_bufs[0], self._dim)
            ops.mala_propose_from_normals(th, self._fresh_grad(), self._z_bufs[0], thp, eps, s2)
            self._pipe_valid, self._unit_ready = True, False
        logu, zt_next = self._take_unit()
        if self._sep_step:
            self._eval_logp(thp, self._lp_p)                                                   # mala.py:46-48, log density
            self._grad_calls += 1   # (the gradient of the same call is evaluated inside the step kernel)
            gp = None
        else:
            gp = self._materialize(self._eval_grad(thp, self._grad_p, self._lp_p), self._grad_p)   # mala.py:46-48
        out = th if self._use_graph else self._new_state()
        with_step = self._prefetch and self.generate_with == "step"
        if self._prefetch and self.serialize_step and self._pf_event is not None and not with_step:
            torch.cuda.current_stream().wait_event(self._pf_event)
        if with_step:
            # both become runnable when the gradient op has finished; the step kernel's launch is handed to
            # the hardware first
            self._ev_ready[self._pf_slot].record(torch.cuda.current_stream())
            if not self.step_first:
                self._start_unit(self._pf_slot, recorded=True)
        if self._sep_step:
            self._model.bk_mala_step(th, out, thp, self._lp, self._lp_p, logu, zt_next, eps, s2,
                                     self._mask, self._ret, self._accepted)                    # mala.py:50-66
        else:
            ops.mala_step(th, out, self._grad, thp, gp, self._lp, self._lp_p, logu, zt_next, eps, s2,
                          self._mask, self._ret, self._accepted)                               # mala.py:50-66
        if with_step and self.step_first:
            self._start_unit(self._pf_slot, recorded=True)
        self._theta_dc = out

    def _draw(self):
        ops = self._ops
        eps = float(self._epsilon)
        th, thp = self._theta_dc, self._theta_p
        if self._prefetch:
            z, logu = self._take_randomness()
            ops.mala_propose_from_normals(th, self._grad, z, thp, eps, math.sqrt(2 * eps))
        elif self._chain_major:
            logu = self._logu
            ops.normals_chain_major(self._rng_kind, self._rng_state, self._zt_bufs[0], self._dim)
            ops.log_uniform(self._rng_kind, self._rng_state, logu)  # right after the normals: same stream order
            ops.mala_propose_from_normals(th, self._grad, self._z_bufs[0], thp, eps, math.sqrt(2 * eps))
        else:
            logu = self._logu
            ops.mala_propose(self._rng_kind, self._rng_state, th, self._grad, thp, eps, math.sqrt(2 * eps))
            ops.log_uniform(self._rng_kind, self._rng_state, logu)  # right after the normals: same stream order
        gp = self._materialize(self._eval_grad(thp, self._grad_p, self._lp_p), self._grad_p)
        ops.mala_logq(th, self._grad, thp, gp, eps, self._fwd, self._rev)
        ops.mh_accept(_lib.ACCEPT_MALA, self._lp, self._fwd, self._lp_p, self._rev, logu,
                      self._mask, self._ret, self._accepted)
        self._select(self._mask, th, thp, self._grad, gp)
