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

import torch

from . import _lib
from ._engine import ManyChainSampler


class HMCDiag(ManyChainSampler):
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
        chain_tile: Optional[int] = None,
        graph: bool = False,
        fuse_builtin: bool = True,
        ops=None,
    ):
        self._stepsize = stepsize
        self._steps = steps
        self._setup(model, metric_diag, init, seed, chains, chain_id0, ops)
        self._chain_tile = self._pick_tile(chain_tile)
        self._init_graph(graph)
        # built-in separable targets can run the whole trajectory in registers
        # (bk_hmc_trajectory_gaussian); results are bit-identical to the step-by-step path
        self._fused = bool(fuse_builtin) and self._batched and hasattr(model, "bk_hmc_trajectory")
        D, C, dev = self._dim, self._C, self._ops.device
        f64 = dict(dtype=torch.float64, device=dev)
        self._rho = torch.empty((D, C), **f64)
        self._theta_p = torch.empty((D, C), **f64)
        self._grad = torch.empty((D, C), **f64)      # gradient at the current point
        self._grad_p = torch.empty((D, C), **f64)    # gradient along / at the end of the trajectory
        self._lp = torch.empty(C, **f64)
        self._lp_p = torch.empty(C, **f64)
        self._kin0 = torch.empty(C, **f64)
        self._kin1 = torch.empty(C, **f64)
        self._logu = torch.empty(C, **f64)
        self._ret = torch.empty(C, **f64)
        self._mask = torch.empty(C, dtype=torch.uint8, device=dev)
        self._accepted = torch.zeros(1, dtype=torch.int32, device=dev)
        self._have_cache = False
        self._draws = 0

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

    # -- one draw for every chain ------------------------------------------------------------------
    def sample(self):
        self._run_draw(self._draw)
        self._draws += 1
        return self._draw_out(self._theta_dc, self._ret)

    def _draw(self):
        ops = self._ops
        eps, L, m = float(self._stepsize), int(self._steps), self._metric_dev
        half = 0.5 * eps
        th, thp, rho = self._theta_dc, self._theta_p, self._rho
        mirror = not self._batched

        # momentum + kinetic energy [hmc.py:56, :37]
        ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, rho, m, self._kin0)

        if self._fused:
            if not self._have_cache:
                self._eval_logp(th, self._lp)
                self._have_cache = True
            self._model.bk_hmc_trajectory(th, thp, rho, rho, m, eps, L)        # [hmc.py:40-53]
            ops.leapfrog_finish(rho, None, None, m, 0.0, False, self._kin1)    # kinetic term [hmc.py:37]
            self._eval_logp(thp, self._lp_p)
            ops.log_uniform(self._rng_kind, self._rng_state, self._logu)
            ops.mh_accept(_lib.ACCEPT_HMC, self._lp, self._kin0, self._lp_p, self._kin1, self._logu,
                          self._mask, self._ret, self._accepted)
            ops.select_columns(self._mask, th, thp)
            return

        if mirror:
            self._eval_logp(th, self._lp)                  # joint_logp(theta, rho)      [hmc.py:57]
            g = self._eval_grad(th, self._grad, None)       # leapfrog's first gradient   [hmc.py:45]
        else:
            if not self._have_cache:
                self._materialize(self._eval_grad(th, self._grad, self._lp), self._grad)
                self._have_cache = True
            g = self._grad

        if L == 0:
            # rho_mid = rho - c*t ; rho1 = rho_mid + c*t ; theta unchanged [hmc.py:46,52]
            ops.kick_drift(th, thp, rho, rho, g, m, 0.0, True, -half, False, 0.0)
            g_last = g
            if not mirror:
                self._lp_p.copy_(self._lp)
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
                        ops.kick_drift(th_t, thp_t, rho_t, rho_t, g_t, m, eps, True, -half, True, eps)
                    else:
                        ops.kick_drift(thp_t, thp_t, rho_t, rho_t, gl, m, eps, False, 0.0, True, eps)
                    want_lp = lp_t if (last and not mirror) else None
                    gl = self._eval_grad(thp_t, gp_t, want_lp)
                if tile:
                    self._materialize(gl, gp_t)
                    g_last = self._grad_p
                else:
                    g_last = gl
        # forward half-step + kinetic energy of the proposal [hmc.py:52, :37]
        ops.leapfrog_finish(rho, None, g_last, m, half, False, self._kin1)
        if mirror:
            self._eval_logp(thp, self._lp_p)                # joint_logp(theta_prop, rho_prop) [hmc.py:59]
        # accept [hmc.py:60-63]
        ops.log_uniform(self._rng_kind, self._rng_state, self._logu)
        ops.mh_accept(_lib.ACCEPT_HMC, self._lp, self._kin0, self._lp_p, self._kin1, self._logu,
                      self._mask, self._ret, self._accepted)
        if mirror:
            ops.select_columns(self._mask, th, thp)
        else:
            gp = self._materialize(g_last, self._grad_p) if L > 0 else None
            if gp is not None:
                ops.select_columns(self._mask, th, thp, self._grad, gp)
            else:
                ops.select_columns(self._mask, th, thp)
