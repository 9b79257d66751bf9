"""Model protocols and type aliases (counterpart of bayes_kit/typing.py:1-42).

Two forms of the reference's structural Model protocol are accepted by the samplers:

* the reference form itself (single chain, NumPy): ``dims()``, ``log_density(theta)``,
  ``log_density_gradient(theta) -> (float, array-like)`` on a length-D float64 vector
  (bayes_kit/typing.py:15-27).  The sampler then runs ONE chain; the integrator, RNG and
  accept still run on the GPU, the model is called on the host once per gradient.
* the batched device form (many chains): the same three method names with a leading chain
  axis, marked by a truthy attribute ``batched``.  ``Theta`` is a float64 device tensor
  view of shape (C, D) with strides (1, ld) over the engine's chain-contiguous buffer;
  ``log_density(Theta) -> (C,)``, ``log_density_gradient(Theta) -> ((C,), (C, D))``.
  Outputs may come back in any strides.
"""
from __future__ import annotations

from typing import Protocol, Sequence, Tuple, Union, runtime_checkable

import numpy as np
from numpy.typing import ArrayLike, NDArray

FloatType = np.float64
IntType = np.int64
VectorType = NDArray[FloatType]
DrawAndLogP = tuple  # (theta, logp)
Seed = Union[int, np.random.BitGenerator, np.random.Generator]
ChainType = Union[Sequence[float], VectorType]


@runtime_checkable
class LogDensityModel(Protocol):
    def dims(self) -> int: ...

    def log_density(self, params_unc): ...


@runtime_checkable
class GradModel(LogDensityModel, Protocol):
    def log_density_gradient(self, params_unc) -> Tuple[float, ArrayLike]: ...


@runtime_checkable
class HessianModel(GradModel, Protocol):
    def log_density_hessian(self, params_unc): ...


@runtime_checkable
class LogPriorLikelihoodModel(LogDensityModel, Protocol):
    def log_prior(self, params_unc): ...

    def log_likelihood(self, params_unc): ...


# ---- the batched device form (what the many-chain engine calls) ---------------------------------
@runtime_checkable
class BatchedLogDensityModel(Protocol):
    """``batched = True`` marks a model that scores all chains in one call."""

    batched: bool

    def dims(self) -> int: ...

    def log_density(self, Theta):
        """(C, D) float64 device tensor (strides (1, ld)) -> (C,) log densities."""
        ...


@runtime_checkable
class BatchedGradModel(BatchedLogDensityModel, Protocol):
    def log_density_gradient(self, Theta):
        """-> ((C,) log densities, (C, D) gradients in any strides)."""
        ...


@runtime_checkable
class EngineTarget(Protocol):
    """Fast path of the library's own targets and of ``CTarget`` plugins: results written in place
    into the engine's [D, n] chain-contiguous buffers (either output may be None)."""

    def bk_eval(self, theta_dc, grad_out, logp_out) -> None: ...


# Optional hooks a model may offer on top of ``bk_eval``; the samplers use the strongest one present (``hasattr``), all give the
# same draws as the plain gradient-op path.  The library's targets and ``CTarget.from_source`` objects provide them; a
# ``TorchModel(compile=True)`` forwards those of the target it was compiled into.
#
#   bk_counted = True, bk_eval(theta, grad, logp, n_dev)   the gradient op takes its chain count from device memory
#                                                           (DrGhmcDiag: lane counts stay on the device, a draw is one hipGraph)
#   bk_leapfrog_step(theta, rho, metric, h, n_dev=None)     one leapfrog step {gradient, kick, drift} as ONE launch, in place
#   bk_mala_step(theta, theta_out, theta_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret, count)
#                          MALA's step kernel with the (separable) density inlined: gradients recomputed, none stored
#   bk_leapfrog_trajectory(theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out, metric, h, steps,
#                          n_dev=None, hmc_first=False)   gathering first step .. last gradient + log density as ONE launch
#                                                           (drghmc.py:280-283, hmc.py:48-50): the step-by-step paths of
#                                                           DrGhmcDiag and HMCDiag issue one launch per step instead of two
#   bk_hmc_trajectory / bk_hmc_draw                         whole HMC trajectory / whole draw of a separable density in registers
#                                                           (hmc.py:40-63)
#   bk_hmc_proposal(theta, rho, grad, theta_out, grad_out, logp_out, kin_out, metric, eps, steps) -> bool
#                                                           whole HMC trajectory of a lane-spread density as ONE launch
#   bk_dr_proposal(...) -> bool, bk_dr_proposal_supported() whole delayed-rejection proposal as ONE launch
#                                                           (drghmc.py:253-289, 319-346, 391-446)
#   bk_gradient(Theta)                                      the gradient alone through PyTorch ops (TorchModel(grad_fn=...))
