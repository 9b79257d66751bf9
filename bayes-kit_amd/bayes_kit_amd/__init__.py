"""bayes_kit_amd -- MI355X-native many-chain engine behind bayes-kit's sampler API.

Drop-in for the hot path of flatironinstitute/bayes-kit (``bayes_kit/__init__.py:15-30``
export list): ``HMCDiag``, ``MALA``, ``DrGhmcDiag`` keep the reference's constructor
signatures and ``sample()`` protocol, and ``rhat`` / ``ess`` / ``iat`` keep its function
signatures, while the arithmetic runs in hand-written HIP kernels for gfx950
(``libbkhip.so``, C ABI in ``include/bkhip.h``).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from . import dist  # noqa: F401
from .diagnostics import DrawRecorder, DrawStore, RunningMoments, rhat_from_moments
# like the reference (bayes_kit/__init__.py:2-13) the function names shadow the sub-modules of
# the same name; importing them through the sub-modules keeps `bayes_kit_amd.rhat` a function
# even after `from bayes_kit_amd.rhat import ...`
from .autocorr import autocorr
from .ess import ess, ess_imse, ess_ipse
from .iat import iat, iat_imse, iat_ipse
from .rhat import rank_normalized_rhat, rhat, split_rhat
from .ensemble import Stretcher
from .drghmc import DrGhmcDiag
from .hmc import HMCDiag
from .mala import MALA
from .metropolis import ChainRng, Metropolis, MetropolisHastings
from .smc import TemperedLikelihoodSMC, TorchPriorLikelihoodModel, hmc_kernel, mala_kernel, metropolis_kernel
from .targets import CTarget, DiagGaussian, Funnel, IsoGaussian, LogisticRegression, TorchModel

__all__ = [
    "DrGhmcDiag",
    "HMCDiag",
    "MALA",
    "Metropolis",
    "MetropolisHastings",
    "TemperedLikelihoodSMC",
    "Stretcher",
    "ess",
    "ess_imse",
    "ess_ipse",
    "iat",
    "iat_imse",
    "iat_ipse",
    "rhat",
    "split_rhat",
    "rank_normalized_rhat",
    "autocorr",
    "RunningMoments",
    "DrawRecorder",
    "DrawStore",
    "rhat_from_moments",
    "IsoGaussian",
    "DiagGaussian",
    "Funnel",
    "LogisticRegression",
    "TorchModel",
    "CTarget",
    "ChainRng",
]
