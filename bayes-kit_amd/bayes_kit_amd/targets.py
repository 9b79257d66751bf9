"""Device-resident models for the many-chain engine.

``IsoGaussian``, ``DiagGaussian`` and ``Funnel`` evaluate their gradient through the
library's C-ABI target entry points (``bk_target_*_grad``, include/bkhip.h) -- the "thin
C-ABI callback" provider of GradModel.log_density_gradient (bayes_kit/typing.py:25-27).
``TorchModel`` adapts any PyTorch-ROCm log-density function through autograd.  All of them
satisfy the batched form of the Model protocol (see typing.py) and can also be called
directly by users.
"""
from __future__ import annotations

import torch

from . import _lib


def _as_dc(Theta):
    """(C, D) chain-major view with strides (1, ld) -> the underlying [D, C] tensor."""
    t = Theta.t()
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        t = Theta.t().contiguous()
    return t


class _BuiltinTarget:
    batched = True
    _kind = ""

    def __init__(self, D, ops=None):
        self._D = int(D)
        self._ops = ops
        self._params = None

    def _get_ops(self):
        if self._ops is None:
            self._ops = _lib.default_ops()
        return self._ops

    def dims(self) -> int:
        return self._D

    # engine fast path: raw [D, n] buffers in, results written in place
    def bk_eval(self, theta_dc, grad_out, logp_out):
        self._get_ops().target_grad(self._kind, self._params, theta_dc, grad_out, logp_out)

    # public batched protocol
    def log_density(self, Theta):
        t = _as_dc(Theta)
        lp = torch.empty(t.shape[1], dtype=torch.float64, device=t.device)
        self.bk_eval(t, None, lp)
        return lp

    def log_density_gradient(self, Theta):
        t = _as_dc(Theta)
        lp = torch.empty(t.shape[1], dtype=torch.float64, device=t.device)
        g = torch.empty_strided(t.shape, (t.stride(0) if t.shape[0] > 1 else t.shape[1], 1),
                                dtype=torch.float64, device=t.device)
        self.bk_eval(t, g, lp)
        return lp, g.t()


class IsoGaussian(_BuiltinTarget):
    """logp = -1/2 theta.theta (BASELINE.json config 2)."""

    _kind = "iso_gaussian"

    def bk_hmc_trajectory(self, theta_in, theta_out, rho_in, rho_out, metric, eps, steps):
        """Whole leapfrog trajectory with the gradient inlined (register-resident)."""
        self._get_ops().hmc_trajectory_gaussian(theta_in, theta_out, rho_in, rho_out, None, metric, eps, steps)


class DiagGaussian(_BuiltinTarget):
    """logp = -1/2 sum_i lam_i theta_i^2 (BASELINE.json config 3)."""

    _kind = "diag_gaussian"

    def __init__(self, lam, ops=None):
        lam_t = torch.as_tensor(lam, dtype=torch.float64)
        super().__init__(lam_t.shape[0], ops)
        self._lam_host = lam_t

    def _lam(self, device):
        if self._params is None or self._params.device != device:
            self._params = self._lam_host.to(device).contiguous()
        return self._params

    def bk_eval(self, theta_dc, grad_out, logp_out):
        self._lam(theta_dc.device)
        super().bk_eval(theta_dc, grad_out, logp_out)

    def bk_hmc_trajectory(self, theta_in, theta_out, rho_in, rho_out, metric, eps, steps):
        """Whole leapfrog trajectory with the gradient inlined (register-resident)."""
        self._get_ops().hmc_trajectory_gaussian(theta_in, theta_out, rho_in, rho_out, self._lam(theta_in.device),
                                                metric, eps, steps)


class Funnel(_BuiltinTarget):
    """Neal's funnel: v = theta_0 ~ N(0, 9), theta_i ~ N(0, e^v) (BASELINE.json config 4)."""

    _kind = "funnel"
    _FUSED_MAX_D = 129  # bk_dr_proposal_funnel keeps a chain's coordinates in one workgroup's registers

    def bk_dr_proposal(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                       kin_out, metric, h, steps):
        """Whole delayed-rejection proposal in one launch; False if the shape is unsupported."""
        if self._D > self._FUSED_MAX_D:
            return False
        self._get_ops().dr_proposal_funnel(theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out,
                                           logp_out, kin_out, metric, h, steps)
        return True


class TorchModel:
    """Any differentiable PyTorch log density, batched over chains.

    ``fn(Theta) -> (C,)`` maps a (C, D) float64 device tensor to per-chain log densities;
    the gradient comes from autograd (one backward of ``lp.sum()``: chains are independent,
    so row c of the result is d lp_c / d theta_c).
    """

    batched = True

    def __init__(self, fn, dims: int):
        self._fn = fn
        self._D = int(dims)

    def dims(self) -> int:
        return self._D

    def log_density(self, Theta):
        with torch.no_grad():
            return self._fn(Theta)

    def log_density_gradient(self, Theta):
        x = Theta.detach().requires_grad_(True)
        with torch.enable_grad():
            lp = self._fn(x)
            (g,) = torch.autograd.grad(lp.sum(), x)
        return lp.detach(), g
