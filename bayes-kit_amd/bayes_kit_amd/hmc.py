"""Many-chain static-trajectory HMC on the GPU: drop-in for ``bayes_kit/hmc.py:8-63``.

Same constructor signature and ``sample() -> (theta, logp)`` / iterator protocol as the
reference's ``HMCDiag``; the momentum draw, leapfrog integrator, Metropolis accept and the
per-chain random streams are HIP kernels behind the C ABI of include/bkhip.h.

Per draw and chain (reference lines in brackets):
  rho ~ N(0, I) from the chain's stream, kin0 = 1/2 rho.(m*rho)            [hmc.py:56, :37]
  back half-step + L x (kick, drift, gradient) + forward half-step       [hmc.py:40-53]
  accept iff log(u) < (lp1 - kin1) - (lp0 - kin0), u always drawn        [hmc.py:57-63]
and the JOINT log density is returned (hmc.py:62-63).

Model calls.  With a reference-style single-chain model the call pattern of the reference is
kept exactly (2 ``log_density`` + ``steps+1`` ``log_density_gradient`` per draw,
test/test_hmc.py:22-35).  With a batched device model the sampler keeps (logp, grad) of the
current point across draws and uses the logp returned with the last gradient of the
trajectory: ``steps`` model calls per draw, bitwise the same results for any model whose
``log_density`` equals the first output of ``log_density_gradient``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib
from ._engine import ManyChainSampler


class HMCDiag(ManyChainSampler):
    TUNING = ("graph", "prefetch_rng", "tune_placement", "chain_tile")
    ENABLE_FUSED_DRAW = True  # experiments / bisection: the one-pass draw kernel for built-in targets
    ENABLE_FUSED_ZT = True    # ... and its chain-major momentum input

    def __init__(
        self,
        model,
        stepsize: float,
        steps: int,
        metric_diag=None,
        init=None,
        seed=None,
        *,
        chains: Optional[int] = None,
        chain_id0: int = 0,
        path: str = "auto",
        metric_dense=None,
        tuning: Optional[dict] = None,
        ops=None,
        **knobs,
    ):
        """The reference's arguments (hmc.py:9-17), then the engine's (ManyChainSampler: chains, chain_id0, path, tuning,
        ops) and ``metric_dense`` (extension: a dense velocity covariance, fp64 MFMA GEMMs).  Tuning knobs (none changes a
        result): ``graph`` (replay a draw as one hipGraph; default: small launch-bound shapes), ``prefetch_rng`` (the next
        draw's randomness on a side stream; default on), ``tune_placement`` (time which allocation plays which role),
        ``chain_tile`` (chains per Infinity-Cache tile)."""
        fuse_builtin, fuse_steps = self._resolve_path(path)
        tn = self._resolve_tuning(tuning, knobs)
        chain_tile, graph = tn.get("chain_tile"), tn.get("graph")
        prefetch_rng, tune_placement = tn.get("prefetch_rng"), tn.get("tune_placement")
        self._stepsize = stepsize
        self._steps = steps
        self._setup(model, metric_diag, init, seed, chains, chain_id0, ops)
        self._chain_tile = self._pick_tile(chain_tile)
        # Dense metric (extension; the reference only has metric_diag, and its literal
        # semantics -- rho ~ N(0, I), kick with m*grad, kinetic rho.(m*rho) -- leave the target
        # invariant only for m = 1).  The dense form is the proper
        # preconditioned HMC in the same (theta, velocity) variables: M = velocity covariance
        # (the inverse mass matrix; ideally ~ the posterior covariance),
        #     rho = chol(M) @ z,  z ~ N(0, I) from the chain's stream
        #     kick  rho += eps * (M @ grad),  drift  theta += eps * rho   [hmc.py:46-52 shape]
        #     kinetic energy 1/2 rho . (M^-1 @ rho)
        # and it reduces to the reference path exactly for M = I.  Every `matrix @ all chains`
        # is one fp64 MFMA GEMM (bk_dense_metric_apply).
        self._M = None
        if metric_dense is not None:
            if metric_diag is not None:
                raise ValueError("give metric_diag or metric_dense, not both")
            if not self._batched:
                raise ValueError("metric_dense needs a batched device model")
            self._install_metric_dense(metric_dense)
            fuse_builtin = False
        self._init_graph(graph, prefetch_rng)
        # built-in separable targets can run the whole trajectory in registers
        # (bk_hmc_trajectory_gaussian); results are bit-identical to the step-by-step path
        self._fused = bool(fuse_builtin) and self._batched and hasattr(model, "bk_hmc_trajectory")
        # ... and, where the target can also sum the energies along the way (bk_hmc_draw_gaussian),
        # a draw is: generator, ONE pass over the state (trajectory + kin0 + kin1 + end-point log
        # density), accept, select.  With Philox streams the momentum is consumed chain-major,
        # straight from the wavefront-per-chain generator: no transpose, no kinetic-energy pass.
        self._fused_draw = self._fused and hasattr(model, "bk_hmc_draw") and self.ENABLE_FUSED_DRAW
        # ... a lane-spread density (bk.Funnel, CTarget.from_source(form="lanes")) runs the whole trajectory -- gradient
        # inlined, theta / rho register-resident, the proposal's gradient, log density and kinetic energy out -- as ONE launch
        # (bk_hmc_proposal: the delayed-rejection proposal kernel with hmc.py's first kick); the step-by-step path otherwise
        # issues ONE launch per leapfrog step where the model has bk_leapfrog_step
        self._lanes_traj = (bool(fuse_builtin) and self._batched and self._M is None and hasattr(model, "bk_hmc_proposal"))
        self._step_hook = (bool(fuse_steps) and self._batched and self._M is None and hasattr(model, "bk_leapfrog_step"))
        # ... and ONE launch per trajectory where it has bk_leapfrog_trajectory (a per-chain density compiled from source:
        # theta in registers, rho in LDS through all L steps), followed by the library's finish launch
        self._traj_hook = (bool(fuse_builtin) and self._step_hook and hasattr(model, "bk_leapfrog_trajectory"))
        self._fused_zt = (self._fused_draw and self._rng_kind == _lib.RNG_PHILOX and self._dim >= 32
                          and self.ENABLE_FUSED_ZT)
        D, C, dev = self._dim, self._C, self._ops.device
        f64 = dict(dtype=torch.float64, device=dev)
        self._rho_bufs = [None if self._fused_zt else torch.empty((D, C), **f64)]
        self._theta_p = torch.empty((D, C), **f64)
        self._grad = torch.empty((D, C), **f64)      # gradient at the current point
        self._grad_p = torch.empty((D, C), **f64)    # gradient along / at the end of the trajectory
        self._lp = torch.empty(C, **f64)
        self._lp_p = torch.empty(C, **f64)
        self._kin0_bufs = [torch.empty(C, **f64)]
        self._kin1 = torch.empty(C, **f64)
        self._logu_bufs = [torch.empty(C, **f64)]
        self._ret = torch.empty(C, **f64)
        self._mask = torch.empty(C, dtype=torch.uint8, device=dev)
        self._accepted = torch.zeros(1, dtype=torch.int32, device=dev)
        if self._fused_draw:
            self._part = torch.empty(12 * C, **f64)    # quarter partials of the three per-chain sums
        if self._fused_zt:
            dp = (D + 7) // 8 * 8
            self._zt_bufs = [torch.empty((C, dp), **f64)]  # (the momentum never exists in the state layout)
        if self._M is not None:
            self._mv = torch.empty((D, C), **f64)      # M @ grad along the trajectory
            self._mv_rng = torch.empty((D, C), **f64)  # M @ rho of the (possibly prefetched) momentum
        self._have_cache = False
        self._draws = 0
        # `self._theta = theta_prop` (hmc.py:61) as a REBIND: with a batched model and eager launches the
        # blend of state and proposal is written to a fresh array that becomes the state and is what sample()
        # returns (never written again): one array written instead of the in-place select's two
        # (bk_blend_columns).  A replayed hipGraph needs fixed addresses and keeps the in-place select.
        self._rebind = self._batched and not self._use_graph
        # Randomness of draw n+1 (momentum, its kinetic energy, the accept uniform) does not
        # depend on draw n, and the reference consumes it in a fixed order (D normals, then one
        # uniform: hmc.py:56,60).  With prefetch_rng it is generated on a second HIP stream
        # while draw n's trajectory streams through HBM on the main one: the RNG kernels are
        # integer bound and a small fraction of a draw, so they hide under the HBM-bound kernels.
        # A hipGraph replay is launch-free already and its graph stays linear (ManyChainSampler: a forked
        # capture is slower and unsafe on this ROCm): with graph=True the randomness is generated in line,
        # whatever prefetch_rng says -- the draws are bit-identical either way.
        if prefetch_rng is None:
            prefetch_rng = self._batched and self._ops.device.type == "cuda"
        self._prefetch = bool(prefetch_rng) and self._batched and not self._use_graph
        self._pf_slot = 0           # double-buffer slot holding the NEXT draw's randomness
        self._pf_ready = False      # ... once it has been generated
        self._pf_event = None       # ... and the event that marks it complete (None: already joined)
        self._pf_kin_stale = False
        self._snap = None           # per slot: the stream table before that slot's normals were generated
        if self._prefetch:
            if self._fused_zt:
                self._zt_bufs.append(torch.empty_like(self._zt_bufs[0]))
                self._rho_bufs.append(None)
            else:
                self._rho_bufs.append(torch.empty((D, C), **f64))
            self._kin0_bufs.append(torch.empty(C, **f64))
            self._logu_bufs.append(torch.empty(C, **f64))
            self._init_side_stream()
            if self._fused_zt:
                self._snap = [torch.empty_like(self._rng_state) for _ in range(2)]
            else:
                self._rng_logical = self._rng_state.clone()  # stream position after the last finished draw
        self.placement = None
        if self._wants_placement_tuning(tune_placement) and not self._fused and self._M is None:
            self._tune_placement()

    def _install_metric_dense(self, metric_dense):
        if isinstance(metric_dense, torch.Tensor) and metric_dense.dim() == 1 and self._M is not None:
            # a DIAGONAL metric given by its entries (the SMC's per-temperature adaptation): factor and inverse are
            # elementwise, formed on the device -- no host round trip, no Cholesky of a D x D matrix
            v = metric_dense.to(device=self._ops.device, dtype=torch.float64)
            if v.shape[0] != self._dim:
                raise ValueError(f"a diagonal metric needs {self._dim} entries")
            self._M.copy_(torch.diag(v))
            self._M_chol.copy_(torch.diag(torch.sqrt(v)))
            self._M_inv.copy_(torch.diag(1.0 / v))
            return
        Mt = torch.as_tensor(metric_dense, dtype=torch.float64).cpu()
        if tuple(Mt.shape) != (self._dim, self._dim):
            raise ValueError(f"metric_dense must be ({self._dim}, {self._dim})")
        Mt = 0.5 * (Mt + Mt.t())
        dev_ = self._ops.device
        chol = torch.linalg.cholesky(Mt)   # host-side set-up
        inv = torch.linalg.inv(Mt)
        inv = 0.5 * (inv + inv.t())
        if self._M is None:
            self._M, self._M_chol, self._M_inv = (x.to(dev_).contiguous() for x in (Mt, chol, inv))
        else:  # in place: launches already queued (or captured) keep pointing at these buffers
            self._M.copy_(Mt)
            self._M_chol.copy_(chol)
            self._M_inv.copy_(inv)

    def set_metric_dense(self, metric_dense):
        """Replace the dense metric of a sampler that was built with one (an adaptation between draws: e.g. the
        tempered SMC re-estimates it from its particles at every temperature); a 1-D tensor = the entries of a diagonal one.  Momenta generated ahead with the old
        metric (prefetch_rng) are discarded: build such samplers with prefetch_rng=False to keep the stream order."""
        if self._M is None:
            raise ValueError("this sampler was built without metric_dense")
        self._install_metric_dense(metric_dense)
        if self._prefetch and self._pf_ready:
            if self._pf_event is not None:
                torch.cuda.current_stream().wait_event(self._pf_event)
            self._pf_ready, self._pf_event = False, None

    def _tune_placement(self):
        """Roles (theta', grad', rho of each slot, grad): see ManyChainSampler._tune_roles."""
        ops, m, eps = self._ops, self._metric_dev, float(self._stepsize)
        n_rho = len(self._rho_bufs)

        builtin = hasattr(self._model, "bk_eval")  # the library's own gradient op: safe to time as well

        def cost(a):
            tp, gp, rhos = a[0], a[1], a[2:2 + n_rho]
            ms = sum(self._time_ms(lambda r=r: ops.kick_drift(tp, tp, r, r, gp, m, eps, False, 0.0, True, eps))
                     for r in rhos) / n_rho
            if builtin:
                ms += self._time_ms(lambda: self._model.bk_eval(tp, gp, None))
            return ms

        chosen, rep = self._tune_roles([self._theta_p, self._grad_p] + list(self._rho_bufs) + [self._grad], cost)
        self._theta_p, self._grad_p = chosen[0], chosen[1]
        self._rho_bufs = chosen[2:2 + n_rho]
        self._grad = chosen[2 + n_rho]
        key = "step_ms" if builtin else "kick_drift_ms"
        self.placement = {key + "_as_allocated": rep["ms_as_allocated"], key + "_chosen": rep["ms_chosen"],
                          "assignments_tried": rep["assignments_tried"]}

    # -- optional cache blocking ----------------------------------------------------------------
    # chain_tile=T runs the L steps tile by tile over blocks of T chains (chains are
    # independent, so this is only a schedule).  The idea: keep a tile's three arrays inside
    # the 256 MiB Infinity Cache.  Measured on MI355X (config 3, T = 8192): the kernels gain
    # ~5 % from the cache but 8x more, 8x smaller launches lose more than that
    # (8.1e7 vs 9.8e7 steps/s), so the default is NO tiling; the knob stays for experiments.
    def _pick_tile(self, chain_tile):
        C = self._C
        if chain_tile is None:
            return C
        t = int(chain_tile)
        return C if t <= 0 or t >= C else max(2, t - t % 2)

    # -- statistics ---------------------------------------------------------------------------
    def accept_rate(self) -> float:
        """Fraction of accepted proposals so far (device counter, ballot/popcount sums)."""
        n = self._draws * self._C
        return float(self._accepted.item()) / n if n else float("nan")

    @property
    def last_accept(self):
        return self._mask.bool() if self._batched else bool(self._mask[0].item())

    def _set_metric(self, m):
        super()._set_metric(m)
        self._pf_kin_stale = True  # a prefetched kinetic energy was computed with the old metric


    def rng_state(self):
        return self._logical_rng().cpu().numpy().view(np.uint64)

    def _state_tensors(self):
        return {"theta": self._theta_dc, "grad": self._grad, "lp": self._lp, "accepted": self._accepted}

    def _logical_rng(self):
        if getattr(self, "_prefetch", False) and self._pf_ready:
            if self._pf_event is not None:
                self._pf_event.synchronize()
            return self._rng_logical if self._snap is None else self._snap[self._pf_slot]
        return self._rng_state

    def load_state_dict(self, sd):
        if self._rebind:
            # the state array is the last draw handed out (see _draw): restore into a fresh one
            self._theta_dc = torch.empty_like(self._theta_dc)
        super().load_state_dict(sd)

    def _after_load(self):
        # drop any randomness generated ahead: it is regenerated from the restored stream
        self._pf_event, self._pf_slot, self._pf_ready, self._pf_kin_stale = None, 0, False, False

    def _refresh_stale_kinetic(self):
        if self._pf_kin_stale and self._pf_ready and not self._fused_draw:
            self._ops.leapfrog_finish(self._rho_bufs[self._pf_slot], None, None, self._metric_dev, 0.0, False,
                                      self._kin0_bufs[self._pf_slot])
        self._pf_kin_stale = False

    def _randomness(self, slot):
        """Momentum, kinetic energy and accept uniform of one draw [hmc.py:56, :37, :60]."""
        ops = self._ops
        if self._fused_zt:
            # D normals, chain-major; their kinetic energy is summed by the draw kernel itself
            # (with prefetch the generator also leaves the table as it found it -- the logical stream
            # position while this slot is the one generated ahead -- in snap[slot])
            ops.normals_chain_major(self._rng_kind, self._rng_state, self._zt_bufs[slot], self._dim,
                                    None if self._snap is None else self._snap[slot])
        elif self._M is None:
            ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, self._rho_bufs[slot],
                                 self._metric_dev, None if self._fused_draw else self._kin0_bufs[slot], None,
                                 self._rng_work)
        else:
            ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, self._mv_rng, None, None,
                                 None, self._rng_work)
            ops.dense_metric_apply(self._M_chol, self._mv_rng, self._rho_bufs[slot])  # rho = chol(M) @ z
            self._dense_kinetic(self._rho_bufs[slot], self._kin0_bufs[slot], self._mv_rng)
        ops.log_uniform(self._rng_kind, self._rng_state, self._logu_bufs[slot])

    def _dense_kinetic(self, rho, kin_out, scratch):
        """kin = 1/2 rho . (M^-1 @ rho)"""
        self._ops.dense_metric_apply(self._M_inv, rho, scratch)
        self._ops.dot_columns(rho, scratch, 0.5, kin_out)

    def _mg(self, g):
        """Gradient as the kick sees it: M @ grad with a dense metric, grad itself otherwise."""
        if self._M is None:
            return g
        self._ops.dense_metric_apply(self._M, self._materialize_dense(g), self._mv)
        return self._mv

    def _materialize_dense(self, g):
        if g.dim() == 2 and g.stride(1) == 1:
            return g
        self._ops.relayout(g, self._grad_p)
        return self._grad_p

    # The side-stream generator (prefetch_rng) runs ONE draw ahead: at the start of draw n the generator
    # of draw n+1 is queued.  `_pf_slot` is the slot the next draw consumes, and the stream position the
    # reference's generator would have between two sample() calls is the table as it was when that
    # slot's generation began: snap[slot], written by the chain-major generator itself, or else
    # _rng_logical (a copy queued in front of the generator).
    # For the one-pass draw (bk_hmc_draw), which is fp64-VALU bound like the generator, two other
    # schedules were measured at 65,536 x 1024 on one box (tools/fused_hmc_profile_run.py): queueing the
    # generator of draw n+2 right after draw n's accept test, so that it starts under the HBM-bound
    # blend -- 1.37 ms per draw, the generator then takes the ALUs from the next draw's trajectories
    # (1.0-1.1 ms instead of 0.78) -- and no side stream at all, 1.38 ms; one ahead gives 1.29 ms: the
    # trajectories keep the ALUs, the generator takes what they leave and finishes under the blend.
    def _current_randomness(self):
        """Buffers holding this draw's randomness (generated ahead with prefetch_rng)."""
        self._zt_slot = 0 if not self._prefetch else self._pf_slot
        if not self._prefetch:
            self._randomness(0)
            return self._rho_bufs[0], self._kin0_bufs[0], self._logu_bufs[0]
        cur = self._pf_slot
        if not self._pf_ready:
            self._randomness(cur)  # very first draw: nothing generated ahead yet
            self._pf_kin_stale = False
        else:
            if self._pf_event is not None:
                torch.cuda.current_stream().wait_event(self._pf_event)
            self._refresh_stale_kinetic()
        return self._rho_bufs[cur], self._kin0_bufs[cur], self._logu_bufs[cur]

    def _start_next_randomness(self):
        """Queue the next draw's generator on the side stream, into the other double-buffer slot (last
        read by the previous draw's kernels, which are already queued on the main stream)."""
        if not self._prefetch:
            return
        main = torch.cuda.current_stream()
        nxt = 1 - self._pf_slot
        ready, ev = self._ev_ready[nxt], self._ev_done[nxt]
        ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            if self._snap is None:
                self._rng_logical.copy_(self._rng_state)
            self._randomness(nxt)
            ev.record(self._side)
        self._pf_event, self._pf_slot, self._pf_ready = ev, nxt, True

    def _graph_key(self):
        return (float(self._stepsize), int(self._steps))

    # -- one draw for every chain ------------------------------------------------------------------
    def sample(self):
        self._run_draw(self._draw)
        self._join_side_stream()
        self._draws += 1
        return self._draw_out(self._theta_dc, self._ret)

    def _draw(self):
        ops = self._ops
        eps, L, m = float(self._stepsize), int(self._steps), self._metric_dev
        half = 0.5 * eps
        th, thp = self._theta_dc, self._theta_p
        mirror = not self._batched

        # momentum + kinetic energy + accept uniform [hmc.py:56, :37, :60]; the uniform is drawn
        # right after the D normals -- the same stream order as the reference, whose uniform is
        # the next value of the stream whatever happens in between
        rho, kin0, logu = self._current_randomness()
        self._start_next_randomness()  # hides under this draw's launches

        if self._fused_draw:
            if not self._have_cache:
                self._eval_logp(th, self._lp)
                self._have_cache = True
            zt = self._zt_bufs[self._zt_slot] if self._fused_zt else None
            # (fp64-VALU bound: a metric of ones is not multiplied in -- x * 1.0 is x, bit for bit)
            m_draw = None if (m is None or self._metric_identity) else m
            self._model.bk_hmc_draw(th, thp, rho, zt, m_draw, eps, L, self._part, kin0, self._kin1, self._lp_p,
                                    accept=(self._lp, logu, self._mask, self._ret, self._accepted))  # [hmc.py:56-63]
            self._take(th, thp)
            return
        if self._fused:
            if not self._have_cache:
                self._eval_logp(th, self._lp)
                self._have_cache = True
            self._model.bk_hmc_trajectory(th, thp, rho, rho, m, eps, L)        # [hmc.py:40-53]
            ops.leapfrog_finish(rho, None, None, m, 0.0, False, self._kin1)    # kinetic term [hmc.py:37]
            self._eval_logp(thp, self._lp_p)
            ops.mh_accept(_lib.ACCEPT_HMC, self._lp, kin0, self._lp_p, self._kin1, logu,
                          self._mask, self._ret, self._accepted)
            self._take(th, thp)
            return

        if self._lanes_traj and L >= 1:
            if not self._have_cache:
                self._materialize(self._eval_grad(th, self._grad, self._lp), self._grad)
                self._have_cache = True
            # [hmc.py:40-53, :59] in one launch; the proposal's gradient is kept for the chains that accept
            if self._model.bk_hmc_proposal(th, rho, self._grad, thp, self._grad_p, self._lp_p, self._kin1, m, eps, L):
                ops.mh_accept(_lib.ACCEPT_HMC, self._lp, kin0, self._lp_p, self._kin1, logu,
                              self._mask, self._ret, self._accepted)                     # [hmc.py:60-63]
                self._take(th, thp, self._grad, self._grad_p)
                return

        if mirror:
            self._eval_logp(th, self._lp)                  # joint_logp(theta, rho)      [hmc.py:57]
            g = self._eval_grad(th, self._grad, None)       # leapfrog's first gradient   [hmc.py:45]
        else:
            if not self._have_cache:
                self._materialize(self._eval_grad(th, self._grad, self._lp), self._grad)
                self._have_cache = True
            g = self._grad

        if self._M is not None:
            m = None  # the metric is applied by _mg()
        if L == 0:
            # rho_mid = rho - c*t ; rho1 = rho_mid + c*t ; theta unchanged [hmc.py:46,52]
            ops.kick_drift(th, thp, rho, rho, self._mg(g), m, 0.0, True, -half, False, 0.0)
            g_last = g
            if not mirror:
                self._lp_p.copy_(self._lp)
        elif (self._traj_hook and not mirror and self._chain_tile >= self._C
              and self._model.bk_leapfrog_trajectory(th, rho, g, None, thp, rho, self._grad_p, self._lp_p, m, eps, L,
                                                     hmc_first=True)):
            self._grad_calls += L   # [hmc.py:45-50] in one launch: L gradients of the model's density
            g_last = self._grad_p
        else:
            g_last = None
            T = self._chain_tile
            for c0 in range(0, self._C, T):
                c1 = min(self._C, c0 + T)
                tile = (c0, c1) != (0, self._C)
                v = (lambda a: a[:, c0:c1]) if tile else (lambda a: a)
                th_t, thp_t, rho_t, gp_t, g_t = v(th), v(thp), v(rho), v(self._grad_p), v(g)
                lp_t = self._lp_p[c0:c1] if tile else self._lp_p
                gl = None
                for n in range(L):
                    last = n == L - 1
                    if n == 0:
                        ops.kick_drift(th_t, thp_t, rho_t, rho_t, self._mg(g_t), m, eps, True, -half, True, eps)
                    elif self._step_hook and not mirror:
                        # {gradient at the point reached, kick, drift} of step n as ONE launch (the gradient op of step
                        # n - 1 and this step's kick+drift, fused: the model's density inside the library's step kernel)
                        self._grad_calls += 1
                        self._model.bk_leapfrog_step(thp_t, rho_t, m, eps)
                    else:
                        ops.kick_drift(thp_t, thp_t, rho_t, rho_t, self._mg(gl), m, eps, False, 0.0, True, eps)
                    if self._step_hook and not mirror and not last:
                        continue  # (the next step's launch evaluates the gradient itself)
                    want_lp = lp_t if (last and not mirror) else None
                    gl = self._eval_grad(thp_t, gp_t, want_lp)
                if tile:
                    self._materialize(gl, gp_t)
                    g_last = self._grad_p
                else:
                    g_last = gl
        # forward half-step + kinetic energy of the proposal [hmc.py:52, :37]
        if self._M is None:
            ops.leapfrog_finish(rho, None, g_last, m, half, False, self._kin1)
        else:
            ops.leapfrog_finish(rho, rho, self._mg(g_last), None, half, False, None)
            self._dense_kinetic(rho, self._kin1, self._mv)
        if mirror:
            self._eval_logp(thp, self._lp_p)                # joint_logp(theta_prop, rho_prop) [hmc.py:59]
        # accept [hmc.py:60-63]
        ops.mh_accept(_lib.ACCEPT_HMC, self._lp, kin0, self._lp_p, self._kin1, logu,
                      self._mask, self._ret, self._accepted)
        if mirror:
            self._select(self._mask, th, thp)
        else:
            gp = self._materialize(g_last, self._grad_p) if L > 0 else None
            self._take(th, thp, self._grad if gp is not None else None, gp)

    def _take(self, th, thp, g=None, gp=None):
        """The accepted chains take their proposal [hmc.py:61] (and its cached gradient)."""
        if not self._rebind:
            self._select(self._mask, th, thp, g, gp)
            return
        self._out = torch.empty_like(th)
        self._ops.blend_columns(self._mask, th, thp, self._out)
        self._theta_dc = self._out
        if gp is not None:
            self._ops.select_columns(self._mask, g, gp)
