"""Device-resident models for the many-chain engine.

``IsoGaussian``, ``DiagGaussian`` and ``Funnel`` evaluate their gradient through the
library's C-ABI target entry points (``bk_target_*_grad``, include/bkhip.h) -- the "thin
C-ABI callback" provider of GradModel.log_density_gradient (bayes_kit/typing.py:25-27).
``CTarget`` loads a user-compiled target with the same calling convention (plugin ABI
``bk_target_fn``).  ``TorchModel`` adapts any PyTorch-ROCm log-density function through autograd.  All of them
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

    # engine fast path: raw [D, n] buffers in, results written in place.  n_dev (optional): the number of chains
    # to evaluate, in device memory (an int32 tensor [1]); the arrays' chain extent is then only a bound --
    # what lets DrGhmcDiag run its data-dependent lane sets without reading their sizes back (bk_counted).
    bk_counted = True

    def bk_eval(self, theta_dc, grad_out, logp_out, n_dev=None):
        if n_dev is None:
            self._get_ops().target_grad(self._kind, self._params, theta_dc, grad_out, logp_out)
        else:
            self._get_ops().target_grad(self._kind, self._params, theta_dc, grad_out, logp_out, n_dev=n_dev)

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


class _GaussianLanes:
    """The separable Gaussians through the lane-spread kernel templates (a separable density is a lanes-form density without
    head coordinates): one launch per delayed-rejection proposal (D <= 128) and one launch per leapfrog step."""

    _FUSED_MAX_D = 128

    def _lam_or_none(self, device):
        return None

    def bk_dr_proposal(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                       kin_out, metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None,
                       ghost=None, ghost0=None):
        """Whole delayed-rejection proposal in one launch (arguments as Funnel.bk_dr_proposal); False if unsupported."""
        if self._D > self._FUSED_MAX_D or max(theta_in.stride(0), theta_out.stride(0)) * 16 * 8 >= 2 ** 32:
            return False
        self._get_ops().dr_proposal_gaussian(self._lam_or_none(theta_in.device), theta_in, rho_in, grad_in, src_index, theta_out,
                                             rho_out, grad_out, logp_out, kin_out, metric, h, steps, n_dev=n_dev,
                                             lanes_out=lanes_out, lanes_total=lanes_total, level=level, job=job, ghost=ghost,
                                             ghost0=ghost0)
        return True

    def bk_dr_proposal_supported(self) -> bool:
        return self._D <= self._FUSED_MAX_D

    def bk_leapfrog_step(self, theta, rho, metric, h, n_dev=None):
        """One leapfrog step {gradient, kick, drift} as ONE launch, theta / rho advanced in place (any D)."""
        self._get_ops().leapfrog_step_gaussian(self._lam_or_none(theta.device), theta, rho, metric, h, n_dev=n_dev)

    def bk_mala_step(self, theta, theta_out, theta_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret, count):
        """MALA's step kernel (proposal densities, accept, select, next proposal; mala.py:41-66) with the density's term inlined:
        both gradients recomputed from theta / theta_prop, none stored (56 D bytes per chain-draw instead of 88 D)."""
        self._get_ops().mala_step_gaussian(self._lam_or_none(theta.device), theta, theta_out, theta_prop, lp, lp_prop, log_u,
                                           zt_next, eps, sqrt2eps, mask, ret, count)


class IsoGaussian(_GaussianLanes, _BuiltinTarget):
    """logp = -1/2 theta.theta (BASELINE.json config 2)."""

    _kind = "iso_gaussian"

    def bk_hmc_trajectory(self, theta_in, theta_out, rho_in, rho_out, metric, eps, steps):
        """Whole leapfrog trajectory with the gradient inlined (register-resident)."""
        self._get_ops().hmc_trajectory_gaussian(theta_in, theta_out, rho_in, rho_out, None, metric, eps, steps)

    def bk_hmc_draw(self, theta_in, theta_out, rho_in, zt, metric, eps, steps, part, kin0, kin1, lp_out, accept=None):
        """Trajectory + energies (+ accept test) of one HMC draw in one pass (bk_hmc_draw_gaussian)."""
        self._get_ops().hmc_draw_gaussian(theta_in, theta_out, rho_in, zt, None, metric, eps, steps, part, kin0, kin1,
                                          lp_out, accept)


class DiagGaussian(_GaussianLanes, _BuiltinTarget):
    """logp = -1/2 sum_i lam_i theta_i^2 (BASELINE.json config 3)."""

    _kind = "diag_gaussian"

    def _lam_or_none(self, device):
        return self._lam(device)

    def __init__(self, lam, ops=None):
        lam_t = torch.as_tensor(lam, dtype=torch.float64)
        super().__init__(lam_t.shape[0], ops)
        self._lam_host = lam_t

    def _lam(self, device):
        if self._params is None or self._params.device != device:
            self._params = self._lam_host.to(device).contiguous()
        return self._params

    def bk_eval(self, theta_dc, grad_out, logp_out, n_dev=None):
        self._lam(theta_dc.device)
        super().bk_eval(theta_dc, grad_out, logp_out, n_dev)

    def bk_hmc_trajectory(self, theta_in, theta_out, rho_in, rho_out, metric, eps, steps):
        """Whole leapfrog trajectory with the gradient inlined (register-resident)."""
        self._get_ops().hmc_trajectory_gaussian(theta_in, theta_out, rho_in, rho_out, self._lam(theta_in.device),
                                                metric, eps, steps)

    def bk_hmc_draw(self, theta_in, theta_out, rho_in, zt, metric, eps, steps, part, kin0, kin1, lp_out, accept=None):
        """Trajectory + energies (+ accept test) of one HMC draw in one pass (bk_hmc_draw_gaussian)."""
        self._get_ops().hmc_draw_gaussian(theta_in, theta_out, rho_in, zt, self._lam(theta_in.device), metric, eps,
                                          steps, part, kin0, kin1, lp_out, accept)


class Funnel(_BuiltinTarget):
    """Neal's funnel: v = theta_0 ~ N(0, 9), theta_i ~ N(0, e^v) (BASELINE.json config 4)."""

    _kind = "funnel"
    _FUSED_MAX_D = 129  # bk_dr_proposal_funnel keeps a chain's coordinates in one workgroup's registers

    def bk_dr_proposal(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                       kin_out, metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None,
                       ghost=None, ghost0=None):
        """Whole delayed-rejection proposal in one launch; False if the shape is unsupported.
        n_dev / lanes_out: device-side lane count in / out; job: a scatter job carried along; ghost: the level's
        accept probability and parent update done by the same launch; ghost0: the produced level's first ghost run
        by the same launch (include/bkhip.h)."""
        if self._D > self._FUSED_MAX_D or max(theta_in.stride(0), theta_out.stride(0)) * 17 * 8 >= 2 ** 32:
            return False
        self._get_ops().dr_proposal_funnel(theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out,
                                           logp_out, kin_out, metric, h, steps, n_dev=n_dev, lanes_out=lanes_out,
                                           lanes_total=lanes_total, level=level, job=job, ghost=ghost,
                                           ghost0=ghost0)
        return True

    def bk_dr_proposal_supported(self) -> bool:
        return self._D <= self._FUSED_MAX_D

    def bk_leapfrog_step(self, theta, rho, metric, h, n_dev=None):
        """One leapfrog step {gradient, kick, drift} (drghmc.py:280-283) as ONE launch, theta / rho advanced in place: what
        the step-by-step paths call instead of {bk_eval, kick_drift} (any D)."""
        self._get_ops().leapfrog_step_funnel(theta, rho, metric, h, n_dev=n_dev)

    def bk_hmc_proposal(self, theta_in, rho, grad_in, theta_out, grad_out, logp_out, kin_out, metric, eps, steps):
        """A whole HMC trajectory (hmc.py:40-53) in ONE launch: theta_out, its gradient and log density, the end point's
        kinetic energy; rho is overwritten.  False if the shape is unsupported (the sampler then steps)."""
        strides = {_lib._ld(t) for t in (theta_in, rho, grad_in, theta_out, grad_out)}
        if self._D > self._FUSED_MAX_D or steps < 1 or len(strides) != 1 or strides.pop() * 17 * 8 >= 2 ** 32:
            return False
        self._get_ops().hmc_trajectory_funnel(theta_in, rho, grad_in, theta_out, grad_out, logp_out, kin_out, metric, eps, steps)
        return True


class LogisticRegression(_BuiltinTarget):
    """Bayesian logistic regression, y_n ~ Bernoulli(sigmoid(x_n . theta)), theta ~ N(0, s^2 I)
    (BASELINE.json config 5; no reference counterpart).  The gradient for ALL chains is two
    fp64 MFMA GEMMs -- Z = X @ Theta (N x C) and G = X^T @ (y - sigmoid(Z)) (D x C) -- with one
    elementwise pass in between; also a batched LogPriorLikelihoodModel (typing.py:37-42) with
    a likelihood temperature for smc.py's annealing."""

    _kind = "logistic"
    bk_counted = False  # (two GEMMs over all chains: no counted form)
    SEGMENTS = 256  # blocks of observations whose log-likelihood partial sums are combined in order

    def __init__(self, X, y, prior_scale: float = 1.0, ops=None):
        X = torch.as_tensor(X, dtype=torch.float64)
        super().__init__(X.shape[1], ops)
        self._N = int(X.shape[0])
        self._X_host, self._y_host = X, torch.as_tensor(y, dtype=torch.float64).reshape(-1)
        self._inv_s2 = 1.0 / float(prior_scale) ** 2
        self._dev = None

    def _buffers(self, device, C, ld):
        """Scratch for C chains whose state rows are `ld` apart (G shares the state's pitch: the finish kernel walks
        theta, G and grad together)."""
        ops = self._get_ops()
        if self._dev is None or self._dev["X"].device != device:
            X = self._X_host.to(device).contiguous()
            self._dev = {"X": X, "Xt": X.t().contiguous(), "y": self._y_host.to(device).contiguous(), "C": 0, "ld": 0}
        b = self._dev
        f64 = dict(dtype=torch.float64, device=device)
        if b["C"] < C:
            b["Z"] = torch.empty((self._N, C), **f64)
            b["part"] = torch.empty((min(self.SEGMENTS, max(1, self._N)), C), **f64)
            b["work"] = ops.gemm_chains_work(self._D, self._N, C)  # split-K slabs of X^T r (a function of D, N only)
            b["C"] = C
        if b["ld"] != ld:
            b["G"] = torch.empty((self._D, ld), **f64)
            b["ld"] = ld
        return b

    def bk_eval(self, theta_dc, grad_out, logp_out, t: float = 1.0, loglik_out=None, gll_out=None):
        """t: likelihood temperature (log density = t * loglik + logprior).  loglik_out ([C]) / gll_out ([D, C], the
        state's pitch): the UNTEMPERED log likelihood and its gradient X^T (y - sigmoid(X theta)) of the evaluated
        points -- what bk_retemper() turns into (logp, grad) at another temperature without touching the data."""
        ops = self._get_ops()
        D, C = theta_dc.shape
        b = self._buffers(theta_dc.device, C, _lib._ld(theta_dc))
        Z = b["Z"][:, :C]
        # (a leapfrog step wants the gradient alone: the residual pass then skips the log likelihood)
        part = b["part"][:, :C] if (logp_out is not None or loglik_out is not None) else None
        if part is None:
            ops.gemm_chains_logistic(b["X"], theta_dc, Z, b["y"])        # r = y - sigmoid(X theta): residual in the epilogue
        else:
            ops.gemm_chains(b["X"], theta_dc, Z)                          # z = X theta          (MFMA)
            ops.logistic_residual(Z, b["y"], part, b["part"].shape[0])  # r = y - sigmoid(z), log-likelihood partials
        G = None
        if grad_out is not None:
            G = b["G"][:, :C]
            ops.gemm_chains(b["Xt"], Z, G, b["work"])     # X^T r                (MFMA, split over N)
        ops.logistic_finish(G, theta_dc, part, self._inv_s2, float(t), grad_out, logp_out, loglik_out)
        if gll_out is not None:
            if G is None:
                raise ValueError("gll_out needs grad_out (the likelihood gradient is formed for it)")
            gll_out.copy_(G)

    def bk_retemper(self, theta_dc, gll, loglik, t: float, grad_out, logp_out):
        """(logp, grad) at temperature t from the untempered parts of an earlier evaluation AT THE SAME POINTS:
        grad = t * gll - theta / s^2, logp = t * loglik - |theta|^2 / (2 s^2).  No pass over the data."""
        self._get_ops().logistic_finish(gll, theta_dc, loglik.reshape(1, -1), self._inv_s2, float(t), grad_out, logp_out,
                                        None)

    # batched LogPriorLikelihoodModel
    def log_likelihood(self, Theta):
        t = _as_dc(Theta)
        ll = torch.empty(t.shape[1], dtype=torch.float64, device=t.device)
        self.bk_eval(t, None, None, 1.0, ll)
        return ll

    def log_prior(self, Theta):
        return -0.5 * self._inv_s2 * (Theta * Theta).sum(dim=1)

    def log_density_gradient_tempered(self, Theta, t: float):
        th = _as_dc(Theta)
        lp = torch.empty(th.shape[1], dtype=torch.float64, device=th.device)
        g = torch.empty_strided(th.shape, (th.stride(0) if th.shape[0] > 1 else th.shape[1], 1),
                                dtype=torch.float64, device=th.device)
        self.bk_eval(th, g, lp, t)
        return lp, g.t()


class CTarget(_BuiltinTarget):
    """A user-compiled device model behind the plugin ABI ``bk_target_fn`` (include/bkhip.h).

    ``library``: path of the user's shared library; ``symbol``: the exported function;
    ``params``: what the function receives as its opaque ``params`` pointer -- a torch tensor
    (its data pointer; device or host), a ``ctypes`` structure / array (host memory, kept alive
    here) or None.  The gradient call goes straight from the sampler to the user's launch: no
    PyTorch ops, no Python callback per chain.

    ``counted_symbol``: optional second export of type ``bk_target_fn_n`` -- the same function taking the
    number of chains from device memory.  With it ``DrGhmcDiag`` keeps the sizes of its lane sets on the
    device and replays a whole delayed-rejection draw as one hipGraph (no host read inside a draw);
    without it the sampler sizes every launch on the host (three reads per draw at K = 3).
    """

    def __init__(self, library: str, symbol: str, dims: int, params=None, ops=None, counted_symbol=None):
        import ctypes

        super().__init__(dims, ops)
        self._cdll = ctypes.CDLL(library)
        I = ctypes.c_int64
        P = ctypes.c_void_p

        def export(name, argtypes):
            try:
                f = getattr(self._cdll, name)
            except AttributeError as e:
                raise _lib.BkHipError(f"{library} does not export {name}") from e
            f.argtypes = argtypes
            f.restype = ctypes.c_int
            return f

        fn = export(symbol, [P, P, P, I, P, I, I, P])
        self._fn, self._symbol = fn, symbol
        self._fn_n = export(counted_symbol, [P, P, P, I, P, I, I, P, P]) if counted_symbol else None
        self._counted_symbol = counted_symbol
        self.bk_counted = self._fn_n is not None
        self._keep = params
        if params is None:
            self._pp = None
        elif isinstance(params, torch.Tensor):
            self._pp = params.data_ptr()
        else:
            self._pp = ctypes.cast(ctypes.pointer(params), P) if not isinstance(params, ctypes.Array) \
                else ctypes.cast(params, P)

    def bk_eval(self, theta_dc, grad_out, logp_out, n_dev=None):
        D, C = theta_dc.shape
        ld = theta_dc.stride(0) if D > 1 else max(C, theta_dc.stride(0))
        if C > 1 and theta_dc.stride(1) != 1:
            raise ValueError("CTarget needs chain-contiguous theta")
        if grad_out is not None and (grad_out.stride(0) if D > 1 else ld) != ld:
            raise ValueError("theta and grad must share their leading dimension")
        stream = torch.cuda.current_stream(theta_dc.device).cuda_stream if theta_dc.is_cuda else None
        args = (theta_dc.data_ptr(), None if grad_out is None else grad_out.data_ptr(),
                None if logp_out is None else logp_out.data_ptr(), ld, self._pp, C, D)
        if n_dev is None:
            rc, name = self._fn(*args, stream), self._symbol
        else:
            if self._fn_n is None:
                raise _lib.BkHipError("this CTarget was loaded without a counted_symbol (bk_target_fn_n)")
            rc, name = self._fn_n(*args, n_dev.data_ptr(), stream), self._counted_symbol
        if rc != 0:
            raise _lib.BkHipError(f"{name} returned {rc}")


# ---------------------------------------------------------------------------------------------------------------
# CTarget.from_source: a density written as a few lines of HIP C++, compiled at construction
# ---------------------------------------------------------------------------------------------------------------
# The translation unit CTarget.from_source generates: the shape as macros, the library's prelude (csrc/bk_source_api.hpp: for
# form="chain" the accessors bk_chain is written against), the user's function, the library's kernels and C entry points around
# it (csrc/bk_source_kernels.hpp; ABI: include/bkhip_source.h).  The kernel text lives in those headers, not here.
_SRC_USER_COMMENT = {
    "elementwise": "// ---- user code: the density is a SUM OVER COORDINATES of bk_term ------------------------------------------------\n//   __device__ void bk_term(double th, i64 d, const double* params, double& term, double& grad)\n//   term = this coordinate's contribution to the log density, grad = d term / d th\n",
    "lanes": '// ---- user code: ONE CHAIN, its coordinates spread over 4 / 8 / 16 lanes of a wavefront (bk_lanes.hpp) ---------------\n//   template <class L> __device__ double bk_lanes_density(L& c, const double* params)\n//   c.dims(), c.head(i), c.sum(f), c.grad_head(i, g), c.grad(f) with f(double theta_d, i64 d); returns the log density\n',
    "chain": '// ---- user code: ONE CHAIN per call ----------------------------------------------------------------------------------\n//   __device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* params)\n//   returns the log density; writes the gradient with g.set(d, value) (a no-op when none is wanted)\n',
}
_SRC_RULE = "// " + "-" * 115 + "\n"

_LANES_MAX_ROWS = 128  # bk_lanes.hpp MAX_ROWS: spread rows a trajectory kernel keeps in registers
_LANES_MAX_HEAD = 8


def _csrc_dir():
    """The library's kernel headers (bk_lanes.hpp, bk_elementwise.hpp, bk_common.hpp): generated sources include them."""
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    for d in (os.path.join(here, "csrc"), os.path.join(here, "..", "csrc")):
        if os.path.exists(os.path.join(d, "bk_common.hpp")):
            return os.path.abspath(d)
    raise _lib.BkHipError("CTarget.from_source: the library's kernel headers (csrc/bk_common.hpp) were not found")


_HIPCC_VERSIONS = {}


def _hipcc_version(hipcc):
    """`hipcc --version` (once per process and compiler path); part of the from_source cache key."""
    if hipcc not in _HIPCC_VERSIONS:
        import subprocess

        try:
            r = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=120)
            _HIPCC_VERSIONS[hipcc] = r.stdout.strip() if r.returncode == 0 else "unknown"
        except (OSError, subprocess.SubprocessError):
            _HIPCC_VERSIONS[hipcc] = "unknown"
    return _HIPCC_VERSIONS[hipcc]


def _check_private(path, what, want_dir):
    """Refuse a cache directory / cached library somebody else could have written: it must be a real directory / regular
    file (no symlink), owned by this user, without group or world write permission."""
    import os
    import stat

    st = os.lstat(path)
    ok_type = stat.S_ISDIR(st.st_mode) if want_dir else stat.S_ISREG(st.st_mode)
    if not ok_type:
        raise _lib.BkHipError(f"CTarget.from_source: {what} {path} is not a plain {'directory' if want_dir else 'file'} "
                              "(a symlink or special file is refused)")
    if st.st_uid != os.getuid():
        raise _lib.BkHipError(f"CTarget.from_source: {what} {path} is owned by uid {st.st_uid}, not by this user "
                              f"({os.getuid()}): refusing to load code from it")
    if st.st_mode & 0o022:
        raise _lib.BkHipError(f"CTarget.from_source: {what} {path} is group- or world-writable "
                              f"(mode {stat.S_IMODE(st.st_mode):o}): refusing to load code from it")


def _source_cache_dir():
    """Where compiled sources are cached: BK_SOURCE_TARGET_DIR, else $XDG_CACHE_HOME/bayes_kit_amd, else
    ~/.cache/bayes_kit_amd -- created 0700 and REFUSED unless it is a real directory owned by this user that nobody else can
    write.  Without a usable home (read-only, unset) a fresh private directory from mkdtemp serves this process."""
    import os
    import tempfile

    explicit = os.environ.get("BK_SOURCE_TARGET_DIR")
    if explicit:
        root = explicit
    else:
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        root = os.path.join(base, "bayes_kit_amd")
    global _PROCESS_CACHE_DIR
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
    except OSError:
        if explicit:
            raise
        if _PROCESS_CACHE_DIR is None:
            _PROCESS_CACHE_DIR = tempfile.mkdtemp(prefix="bayes_kit_amd_src_")  # (0700, unique: nobody can have planted it)
        return _PROCESS_CACHE_DIR
    _check_private(root, "the cache directory", True)
    try:
        _check_ancestors(root)
    except _lib.BkHipError as e:
        if explicit:
            raise
        # the default location sits below a directory somebody else can write: do not trust what is cached there, and do
        # not fail either -- a fresh private directory serves this process (nothing is reused across processes then)
        import warnings

        warnings.warn(f"{e}; compiling into a private temporary directory instead (no cache across processes)", stacklevel=3)
        if _PROCESS_CACHE_DIR is None:
            _PROCESS_CACHE_DIR = tempfile.mkdtemp(prefix="bayes_kit_amd_src_")
        return _PROCESS_CACHE_DIR
    return root


def _check_ancestors(path):
    """The directories ABOVE the cache directory: whoever can rename or replace one of them can swap the whole cache under a
    running process (a check of the leaf alone leaves that window).  Every ancestor must belong to this user or to root and
    must not be writable by group or others -- unless it is sticky (/tmp-like: others cannot rename or delete an entry they
    do not own, and the entry below it has just been checked to be ours)."""
    import os
    import stat

    me = os.getuid()
    p = os.path.realpath(path)
    while True:
        parent = os.path.dirname(p)
        if parent == p:
            return
        st = os.stat(parent)
        if st.st_uid not in (0, me):
            raise _lib.BkHipError(f"CTarget.from_source: {parent}, a directory above the cache directory {path}, is owned by uid "
                                  f"{st.st_uid} (neither this user nor root): refusing to load code from below it")
        if (st.st_mode & 0o022) and not (st.st_mode & stat.S_ISVTX):
            raise _lib.BkHipError(f"CTarget.from_source: {parent}, a directory above the cache directory {path}, is group- or "
                                  f"world-writable without the sticky bit (mode {stat.S_IMODE(st.st_mode):o}): whoever can write "
                                  "it can replace the cache; refusing to load code from below it")
        p = parent


_PROCESS_CACHE_DIR = None


def _find_hipcc():
    import os
    import shutil

    for c in (os.environ.get("HIPCC"), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")):
        if c and os.path.exists(c):
            return c
    c = shutil.which("hipcc")
    if c:
        return c
    raise _lib.BkHipError("CTarget.from_source compiles its source with hipcc when the object is built, and no hipcc was "
                          "found (HIPCC, $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc, PATH).  Build the target on a box with "
                          "ROCm and load the cached library there, or use CTarget(library, symbol, ...) with a library you "
                          "compiled yourself.")


def _source_text(user_source: str, form: str, dims: int, head: int, stage: str = "auto") -> str:
    if form not in ("elementwise", "chain", "lanes"):
        raise ValueError("form must be 'elementwise' (bk_term: one coordinate's term and derivative), 'chain' "
                         "(bk_chain: one chain's log density and gradient, one lane per chain) or 'lanes' "
                         "(bk_lanes_density: one chain spread over the lanes of a wavefront)")
    if stage not in ("auto", "registers", "lds"):
        raise ValueError("stage must be 'auto', 'registers' or 'lds'")
    if stage != "auto" and form != "chain":
        raise ValueError("stage= applies to form='chain' only (where the kernels stage a chain's coordinates before the call)")
    if (stage == "registers" and int(dims) > 128) or (stage == "lds" and int(dims) > 300):
        raise ValueError("stage='registers' needs dims <= 128, stage='lds' dims <= 300 (the one-launch step and trajectory kernels: "
                         "dims <= 128)")
    if form == "lanes":
        if not 0 <= int(head) <= _LANES_MAX_HEAD:
            raise ValueError(f"head must be 0..{_LANES_MAX_HEAD} (the coordinates every lane of a chain holds)")
        if dims < max(1, head):
            raise ValueError("dims must be at least max(1, head)")
        rows = dims - head
        shape = {"HEAD": int(head), "SL": 0 if rows > _LANES_MAX_ROWS else max(1, -(-rows // 16))}
    elif form == "elementwise":
        shape = {"SL": 0 if dims > _LANES_MAX_ROWS else max(1, -(-dims // 16))}  # (slots per class; 0: no lane-spread kernels)
    else:
        # STAGE = D when a chain's coordinates fit the registers of its lane (D <= 128), LDS = D when 64 chains' coordinates fit
        # a workgroup's LDS instead (D <= 300)
        if stage == "lds" and int(dims) <= 128:
            # theta in LDS in every kernel (the step and trajectory kernels too): for long functions, see bk_source_kernels.hpp
            shape = {"DIMS": int(dims), "STAGE": 0, "LDS": int(dims), "THETA_LDS": 1}
        else:
            shape = {"DIMS": int(dims), "STAGE": int(dims) if int(dims) <= 128 else 0,
                     "LDS": int(dims) if 128 < int(dims) <= 300 else 0}
    defines = "".join(f"#define BK_SOURCE_{k} {v}\n" for k, v in shape.items())
    return (f"// Generated by bayes_kit_amd.CTarget.from_source(form=\"{form}\"): plugin ABI bk_target_fn / bk_target_fn_n "
            "(include/bkhip.h) + the entry points of include/bkhip_source.h.\n"
            f"#define BK_SOURCE_FORM_{form} 1\n" + defines + "#include \"bk_source_api.hpp\"\n"
            + _SRC_USER_COMMENT[form] + user_source.rstrip("\n") + "\n" + _SRC_RULE + "#include \"bk_source_kernels.hpp\"\n")


_SRC_REQUIRED_EXPORTS = ("bk_src_target", "bk_src_target_n")


def _compile_source_target(user_source: str, form: str, contract: bool, dims: int = 1, head: int = 0, stage: str = "auto") -> str:
    """hipcc the generated translation unit into a shared library (cached by content: the generated text, the flags and the
    library headers it includes); returns its path."""
    import hashlib
    import os
    import subprocess

    text = _source_text(user_source, form, int(dims), int(head), stage)
    rec = os.environ.get("BK_SOURCE_RECORD")
    if rec:   # (a log of what was asked for, one JSON object per line: `prewarm_sources` builds such a list in parallel)
        import json

        with open(rec, "a") as f:
            f.write(json.dumps(dict(user_source=user_source, form=form, contract=bool(contract), dims=int(dims),
                                    head=int(head), stage=stage)) + "\n")
    csrc = _csrc_dir()
    inc = os.path.abspath(os.path.join(csrc, "..", "..", "include"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
             "-ffp-contract=" + ("fast" if contract else "off")]
    if form == "chain" and int(dims) <= 300:
        # (the staged paths of k_src_chain want the user's loops over d unrolled completely: the private copy of the chain's
        # coordinates then lives in registers, LDS reads are issued in batches; clang's default budget stops at trip counts of ~60)
        flags += ["-mllvm", "-unroll-threshold=%d" % (4000 if int(dims) <= 128 else 10000)]
    # the cache key: the generated text, the flags, every library header the translation unit can include, and the
    # COMPILER (path + `--version` text: a toolchain upgrade must not serve a library built by the old one)
    hipcc = _find_hipcc()
    h = hashlib.sha256((text + " ".join(flags) + "\0" + hipcc + "\0" + _hipcc_version(hipcc)).encode())
    for name in ("bk_common.hpp", "bk_lanes.hpp", "bk_elementwise.hpp", "bk_mala_step.hpp", "bk_source_api.hpp", "bk_source_kernels.hpp",
                 os.path.join(inc, "bkhip.h"), os.path.join(inc, "bkhip_source.h"), os.path.join(inc, "bkhip_math.h")):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    tag = h.hexdigest()[:20]
    root = _source_cache_dir()
    lib = os.path.join(root, f"libt_{tag}.so")
    if os.path.lexists(lib):
        _check_private(lib, "the cached library", False)
        return lib
    # (several ranks -- or several threads of prewarm_sources: two dims of one D-independent source -- may compile the same
    # text at once: everything is written under per-process, per-thread names, the finished library is checked for its
    # exports and only then published with an atomic rename)
    import threading

    me = f"{os.getpid()}.{threading.get_ident():x}"
    src = os.path.join(root, f"t_{tag}.{me}.hip")
    tmp = lib + f".{me}.tmp"
    try:
        fd = os.open(src, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "w") as f:
            f.write(text)
        r = subprocess.run([hipcc] + flags + ["-I" + csrc, "-I" + inc, src, "-o", tmp], capture_output=True, text=True)
        if r.returncode != 0:
            raise _lib.BkHipError("CTarget.from_source: hipcc failed\n" + r.stderr[-6000:])
        os.chmod(tmp, 0o700)
        # (a build that was cut short must not be published: the export names have to be in the library's string table;
        # checked on the bytes -- loading the temporary file would register its code objects a second time)
        with open(tmp, "rb") as f:
            blob = f.read()
        for name in _SRC_REQUIRED_EXPORTS:
            if b"\0" + name.encode() + b"\0" not in blob:
                raise _lib.BkHipError(f"CTarget.from_source: the compiled library does not export {name}")
        os.replace(tmp, lib)
    finally:
        for f_ in (src, tmp):
            try:
                os.unlink(f_)
            except OSError:
                pass
    return lib


def prewarm_sources(specs, workers=None, errors=None):
    """Build the libraries of many from_source densities AHEAD, in parallel (one hipcc each, `workers` at a time): a job of
    several models -- or a test-suite -- then finds them in the cache instead of compiling them one after the other at
    construction.  specs: dicts with the arguments of a from_source call as `BK_SOURCE_RECORD=<file>` logs them
    (user_source, form, contract, dims, head, stage).  Returns (built or found, failed); `errors`: a list that receives
    (form, dims, message tail) of every failed build."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    seen, todo = set(), []
    for sp in specs:
        key = (sp["user_source"], sp["form"], bool(sp.get("contract", False)), int(sp["dims"]), int(sp.get("head", 0)),
               sp.get("stage", "auto"))
        if key not in seen:
            seen.add(key)
            todo.append(key)
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            workers = os.cpu_count() or 1
    ok = bad = 0

    def one(key):
        try:
            _compile_source_target(*key)
            return True
        except _lib.BkHipError as e:
            if errors is not None:
                errors.append((key[1], key[3], str(e)[-400:]))
            return False

    with ThreadPoolExecutor(max_workers=max(1, min(int(workers), 32))) as ex:
        for good in ex.map(one, todo):
            ok, bad = ok + bool(good), bad + (not good)
    return ok, bad


def _bind_source_fast_paths(t):
    """Attach the whole-proposal / whole-draw hooks the samplers look for (bk_dr_proposal, bk_hmc_draw,
    bk_hmc_trajectory) when the generated library exports them: the same hooks the built-in targets have."""
    import ctypes
    import types

    I, P, F = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
    ptr = _lib.ptr

    def export(name, argtypes):
        try:
            f = getattr(t._cdll, name)
        except AttributeError:
            return None
        f.argtypes, f.restype = argtypes, ctypes.c_int
        return f

    def check(rc, name):
        if rc != 0:
            raise _lib.BkHipError(f"{name} returned {rc}")

    def stream(x):
        return torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else None

    f_traj = export("bk_src_hmc_trajectory", [P, P, P, P, I, P, P, F, I, I, I, P])
    f_draw = export("bk_src_hmc_draw", [P, P, I, P, P, I, P, P, F, I, P, P, P, P, P, P, P, P, P, I, I, P])
    f_prop = export("bk_src_dr_proposal_job", [P, P, P, I, P, P, P, P, P, P, I, P, F, I, I, I, P, P, P, P, P, P, P, P, P, P, P])

    f_mala = export("bk_src_mala_step", [P, P, P, I, P, P, P, P, P, I, F, F, P, P, P, I, I, P])
    if f_mala is not None:
        def bk_mala_step(self, theta, theta_out, theta_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret, count):
            """MALA's step kernel with the compiled term inlined (arguments as the built-in Gaussians' bk_mala_step)."""
            D, C = theta.shape
            ld = _lib._ld(theta)
            assert _lib._ld(theta_out) == ld and _lib._ld(theta_prop) == ld
            ldz = 0
            if zt_next is not None:
                assert zt_next.shape[0] == C and zt_next.stride(1) == 1 and zt_next.shape[1] >= D
                ldz = zt_next.stride(0)
            check(f_mala(ptr(theta), ptr(theta_out), ptr(theta_prop), ld, self._pp, ptr(lp), ptr(lp_prop), ptr(log_u),
                         ptr(zt_next), ldz, eps, sqrt2eps, ptr(mask), ptr(ret), ptr(count), C, D, stream(theta)),
                  "bk_src_mala_step")

        t.bk_mala_step = types.MethodType(bk_mala_step, t)
    f_step = export("bk_src_leapfrog_step", [P, P, I, P, F, P, I, I, P, P])
    if f_step is not None:
        def bk_leapfrog_step(self, theta, rho, metric, h, n_dev=None):
            """One leapfrog step {gradient, kick, drift} (drghmc.py:280-283) of the chains in theta / rho ([D, n], advanced in
            place) as ONE launch with the compiled density inlined; n_dev: the lane count in device memory."""
            D, n = theta.shape
            ld = _lib._ld(theta)
            assert _lib._ld(rho) == ld
            check(f_step(ptr(theta), ptr(rho), ld, ptr(metric), h, self._pp, n, D, ptr(n_dev), stream(theta)),
                  "bk_src_leapfrog_step")

        t.bk_leapfrog_step = types.MethodType(bk_leapfrog_step, t)
    f_ctraj = export("bk_src_trajectory", [P, P, P, I, P, P, P, P, P, I, P, F, I, I, I, P, ctypes.c_int, P, P])
    if f_ctraj is not None:
        def bk_leapfrog_trajectory(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out, metric, h,
                                   steps, n_dev=None, hmc_first=False):
            """The whole trajectory of a proposal -- gathering first step, (steps - 1) x {gradient, kick, drift}, last gradient +
            log density (drghmc.py:276-285; hmc.py:46-50 with hmc_first) -- as ONE launch with the compiled per-chain density
            inlined; the caller's leapfrog_finish does the last half-kick and the energies.  False if the shape is unsupported."""
            D, n = theta_out.shape
            ld_in, ld_out = _lib._ld(theta_in), _lib._ld(theta_out)
            if steps < 1 or _lib._ld(rho_in) != ld_in or _lib._ld(grad_in) != ld_in or _lib._ld(rho_out) != ld_out \
                    or _lib._ld(grad_out) != ld_out:
                return False
            check(f_ctraj(ptr(theta_in), ptr(rho_in), ptr(grad_in), ld_in, ptr(src_index), ptr(theta_out), ptr(rho_out),
                          ptr(grad_out), ptr(logp_out), ld_out, ptr(metric), h, steps, n, D, ptr(n_dev), 1 if hmc_first else 0,
                          self._pp, stream(theta_in)), "bk_src_trajectory")
            return True

        t.bk_leapfrog_trajectory = types.MethodType(bk_leapfrog_trajectory, t)
    f_hmcl = export("bk_src_hmc_trajectory_lanes", [P, P, P, P, P, P, P, I, P, F, I, I, I, P, P])
    if f_hmcl is not None:
        def bk_hmc_proposal(self, theta_in, rho, grad_in, theta_out, grad_out, logp_out, kin_out, metric, eps, steps):
            """A whole HMC trajectory (hmc.py:40-53) in ONE launch with the compiled density inlined (arguments as
            Funnel.bk_hmc_proposal); False if the shape is unsupported."""
            D, C = theta_in.shape
            strides = {_lib._ld(x) for x in (theta_in, rho, grad_in, theta_out, grad_out)}
            if steps < 1 or len(strides) != 1:
                return False
            ld = strides.pop()
            if ld * (16 + self._head) * 8 >= 2 ** 32:
                return False
            check(f_hmcl(ptr(theta_in), ptr(rho), ptr(grad_in), ptr(theta_out), ptr(grad_out), ptr(logp_out), ptr(kin_out), ld,
                         ptr(metric), eps, steps, C, D, self._pp, stream(theta_in)), "bk_src_hmc_trajectory_lanes")
            return True

        t.bk_hmc_proposal = types.MethodType(bk_hmc_proposal, t)
    if f_traj is not None:
        def bk_hmc_trajectory(self, theta_in, theta_out, rho_in, rho_out, metric, eps, steps):
            """Whole leapfrog trajectory with the compiled term inlined (register-resident; bk_elementwise.hpp)."""
            D, C = theta_in.shape
            ld = _lib._ld(theta_in)
            assert _lib._ld(theta_out) == ld and _lib._ld(rho_in) == ld and _lib._ld(rho_out) == ld
            check(f_traj(ptr(theta_in), ptr(theta_out), ptr(rho_in), ptr(rho_out), ld, self._pp, ptr(metric), eps, steps,
                         C, D, stream(theta_in)), "bk_src_hmc_trajectory")

        t.bk_hmc_trajectory = types.MethodType(bk_hmc_trajectory, t)
    if f_draw is not None:
        def bk_hmc_draw(self, theta_in, theta_out, rho_in, zt, metric, eps, steps, part, kin0, kin1, lp_out, accept=None):
            """Trajectory + energies (+ accept test) of one HMC draw in one pass, the compiled term inlined."""
            D, C = theta_in.shape
            ld = _lib._ld(theta_in)
            assert _lib._ld(theta_out) == ld and (rho_in is None or _lib._ld(rho_in) == ld) and part.numel() >= 12 * C
            ldz = 0
            if zt is not None:
                assert zt.shape[0] == C and zt.stride(1) == 1 and zt.shape[1] >= D
                ldz = zt.stride(0)
            lp_cur, log_u, mask, ret, count = accept if accept is not None else (None,) * 5
            check(f_draw(ptr(theta_in), ptr(theta_out), ld, ptr(rho_in), ptr(zt), ldz, self._pp, ptr(metric), eps, steps,
                         ptr(part), ptr(kin0), ptr(kin1), ptr(lp_out), ptr(lp_cur), ptr(log_u), ptr(mask), ptr(ret),
                         ptr(count), C, D, stream(theta_in)), "bk_src_hmc_draw")

        t.bk_hmc_draw = types.MethodType(bk_hmc_draw, t)
    if f_prop is not None:
        def bk_dr_proposal(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out, kin_out,
                           metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None, ghost=None,
                           ghost0=None):
            """Whole delayed-rejection proposal in one launch with the compiled density inlined (bk_lanes.hpp;
            arguments as Funnel.bk_dr_proposal); False if the shape is unsupported."""
            D, n = theta_out.shape
            ld_in, ld_out = _lib._ld(theta_in), _lib._ld(theta_out)
            if max(ld_in, ld_out) * (16 + self._head) * 8 >= 2 ** 32:
                return False
            assert _lib._ld(rho_in) == ld_in and (grad_in is None or _lib._ld(grad_in) == ld_in)
            assert _lib._ld(rho_out) == ld_out and (grad_out is None or _lib._ld(grad_out) == ld_out)
            H, hh, live = level if level is not None else (None, None, None)
            check(f_prop(ptr(theta_in), ptr(rho_in), ptr(grad_in), ld_in, ptr(src_index), ptr(theta_out), ptr(rho_out),
                         ptr(grad_out), ptr(logp_out), ptr(kin_out), ld_out, ptr(metric), h, steps, n, D, ptr(n_dev),
                         ptr(lanes_out), ptr(lanes_total), ptr(H), ptr(hh), ptr(live),
                         None if job is None else ctypes.cast(ctypes.pointer(job), P),
                         None if ghost is None else ctypes.cast(ctypes.pointer(ghost), P),
                         None if ghost0 is None else ctypes.cast(ctypes.pointer(ghost0), P), self._pp,
                         stream(theta_in)), "bk_src_dr_proposal_job")
            return True

        t.bk_dr_proposal = types.MethodType(bk_dr_proposal, t)
        t.bk_dr_proposal_supported = lambda: True


def _ctarget_from_source(cls, source: str, dims: int, params=None, form: str = "elementwise", contract: bool = False,
                         ops=None, head: int = 0, stage: str = "auto"):
    """A device model from a few lines of HIP C++, compiled with hipcc when the object is built (cached by content
    in a private per-user directory: $XDG_CACHE_HOME/bayes_kit_amd or ~/.cache/bayes_kit_amd, BK_SOURCE_TARGET_DIR
    overrides; a directory or cached library that another user could have written is refused) into the plugin ABI --
    both forms, so every sampler, DrGhmcDiag's device-side lane counts and hipGraph replay included, treats it like a
    built-in target: the gradient call goes from the sampler straight to the compiled launch, no PyTorch ops, no build to
    write.  The integrator text stays the library's (csrc/bk_elementwise.hpp, csrc/bk_lanes.hpp): the compiled function
    is the only inlined callee.

    form="elementwise": the log density is a sum over coordinates; ``source`` defines
        ``__device__ void bk_term(double th, i64 d, const double* params, double& term, double& grad)``
    and the library supplies the kernels: a streaming 16-byte-per-lane gradient kernel, per-chain sums in the library's
    own order, and the register-resident whole-trajectory / whole-draw HMC kernels the built-in Gaussians have
    (``HMCDiag`` then runs a draw as generator + ONE pass over the state, bit-identical to the step-by-step path).
    form="lanes": a density of head coordinates and sums over the other ("spread") rows -- hierarchical models;
    ``head`` = number of leading coordinates every lane holds; ``source`` defines
        ``template <class L> __device__ double bk_lanes_density(L& c, const double* params)``
    with ``c.dims()``, ``c.head(i)``, ``c.sum(f)``, ``c.grad_head(i, g)``, ``c.grad(f)`` (``f(double theta_d, i64 d)``).  A
    chain is served by 4 / 8 / 16 lanes of a wavefront, sums reduced by DPP in a fixed order; with dims - head <= 128
    ``DrGhmcDiag`` runs every delayed-rejection proposal as ONE launch (the path ``bk.Funnel`` has), else the counted
    step-by-step path.  form="chain": any density; ``source`` defines
        ``__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* params)``
    called by one lane per chain (``th[d]``, ``g.set(d, v)``); the chain's coordinates are staged before the call -- in the
    lane's registers for dims <= 128 (the function's loops over d are then unrolled completely), in LDS for dims <= 300; ``stage="lds"``
    stages them in LDS for dims <= 128 too, in the one-launch step and trajectory kernels as well: for LONG functions (many passes,
    transcendentals, constants) whose unrolled form would not fit a lane's registers beside its coordinates -- their loops may then
    stay rolled.  ``params``: a float64 device tensor (or None).
    **Contract of ``bk_chain``: whenever ``g.wanted()`` it calls ``g.set(d, v)`` for EVERY d in [0, D), exactly once each,
    with the final value** -- in the one-launch step and trajectory kernels ``set`` is not a store but the kick (and
    drift) of coordinate d, so a second ``set`` of the same d kicks twice and a skipped d is left un-kicked; the function
    must also not read back what it has set.  When the object is built on a box with a GPU the two kernels are checked
    against {gradient op, library kick + drift} on random points and are DROPPED (with a warning; the samplers then run the
    gradient as a separate op) if they disagree: ``chain_hooks_note`` says what happened.
    contract=False compiles with -ffp-contract=off (every product and sum rounded, as NumPy does)."""
    lib = _compile_source_target(source, form, contract, dims, head, stage)
    if params is not None and not (isinstance(params, torch.Tensor) and params.dtype == torch.float64):
        raise TypeError("params must be a float64 torch tensor (device memory the compiled function reads) or None")
    t = cls(lib, "bk_src_target", dims, params=params, ops=ops, counted_symbol="bk_src_target_n")
    t.source_library = lib
    t.source_form = form
    t._head = int(head)
    _bind_source_fast_paths(t)
    t.chain_hooks_note = _check_chain_hooks(t) if form == "chain" else None
    return t


def _check_chain_hooks(t, chains: int = 70, h: float = 0.0137):
    """The one-launch leapfrog step and trajectory of a per-chain density deliver every gradient entry INTO the integrator
    (csrc/bk_source_api.hpp: BkGrad::set is a kick, not a store): a bk_chain that sets an entry twice, or skips one, gives
    silently wrong trajectories there while its gradient op looks fine.  Run both kernels once against the composition
    {gradient op, the library's kick + drift arithmetic} on random points; any difference drops the hooks."""
    import warnings

    hooks = [n for n in ("bk_leapfrog_step", "bk_leapfrog_trajectory") if hasattr(t, n)]
    if not hooks:
        return "no one-launch kernels for this shape"
    if not torch.cuda.is_available():
        return "not checked (no GPU where the object was built)"
    D, C = t.dims(), chains
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(20246)
    rnd = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dev)  # noqa: E731
    metric = (torch.rand(D, generator=g, dtype=torch.float64) + 0.5).to(dev)

    def grad_of(theta, want_lp=False):
        gr = torch.empty_like(theta)
        lp = torch.empty(theta.shape[1], dtype=torch.float64, device=dev) if want_lp else None
        t.bk_eval(theta, gr, lp)
        return gr, lp

    def kick_drift(theta, rho, gr, hk):   # rho + hk * (metric * grad), then theta + h * rho: every product and sum rounded
        r = rho + hk * (metric[:, None] * gr)
        return theta + h * r, r

    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=1.25e300), torch.nan_to_num(b, nan=1.25e300))  # noqa: E731
    bad = []
    try:
        theta, rho = rnd(D, C, scale=0.4), rnd(D, C)
        g0, _ = grad_of(theta)
        if not bool(torch.isfinite(g0).any()):
            return "not checked (the gradient is not finite at the random test points)"
        if "bk_leapfrog_step" in hooks:
            th1, r1 = theta.clone(), rho.clone()
            t.bk_leapfrog_step(th1, r1, metric, h)
            th_ref, r_ref = kick_drift(theta, rho, g0, h)
            if not (same(th1, th_ref) and same(r1, r_ref)):
                bad.append("bk_leapfrog_step")
        if "bk_leapfrog_trajectory" in hooks:
            tho, ro, go = torch.empty_like(theta), torch.empty_like(theta), torch.empty_like(theta)
            lpo = torch.empty(C, dtype=torch.float64, device=dev)
            if t.bk_leapfrog_trajectory(theta, rho, g0, None, tho, ro, go, lpo, metric, h, 2) is not False:
                th_a, r_a = kick_drift(theta, rho, g0, 0.5 * h)   # drghmc.py:276-278
                g1, _ = grad_of(th_a)
                th_b, r_b = kick_drift(th_a, r_a, g1, h)          # drghmc.py:280-283
                g2, lp2 = grad_of(th_b, want_lp=True)
                if not (same(tho, th_b) and same(ro, r_b) and same(go, g2) and same(lpo, lp2)):
                    bad.append("bk_leapfrog_trajectory")
        torch.cuda.synchronize()
    except _lib.BkHipError as e:
        return f"not checked ({e})"
    if not bad:
        return "checked: one-launch step / trajectory == gradient op + kick + drift on random points"
    for name in hooks:   # (a function that breaks the contract in one kernel is not trusted in the other)
        delattr(t, name)
    note = ("dropped " + ", ".join(hooks) + ": " + ", ".join(bad) + " differ(s) from {gradient op, kick + drift} -- bk_chain must "
            "call g.set(d, v) exactly once for every d, with the final value, and not read it back (CTarget.from_source docstring)")
    warnings.warn("CTarget.from_source(form='chain'): " + note + "; the samplers will run the gradient as a separate op",
                  stacklevel=3)
    return note


CTarget.from_source = classmethod(_ctarget_from_source)


class TorchModel:
    """Any differentiable PyTorch log density, batched over chains.

    ``fn(Theta) -> (C,)`` maps a (C, D) float64 device tensor to per-chain log densities;
    the gradient comes from autograd (one backward of ``lp.sum()``: chains are independent,
    so row c of the result is d lp_c / d theta_c).

    layout="dc": ``fn`` is written for the engine's own layout instead -- it receives the (D, C) array the
    samplers hold (chains contiguous, e.g. ``-0.5 * (Th * Th * lam[:, None]).sum(dim=0)``).  PyTorch then runs its
    contiguous, vectorised elementwise kernels instead of the generic strided ones the transposed (C, D) view of
    the same memory gets, and autograd returns the gradient in the layout the streamed kick + drift kernel reads
    (no turn through LDS): config-3 shape 2.6 -> 1.8 ms per leapfrog step.  Same values either way.

    grad_fn (optional): the gradient written out in PyTorch ops, same argument and the argument's shape back.  The
    samplers then never build an autograd graph, and a leapfrog step -- which discards the log density (hmc.py:45,50)
    -- calls ``grad_fn`` alone: for an elementwise density that is one torch kernel instead of autograd's eight passes.

    compile=True: an autograd-free fast lane.  ``fn`` is read once with ``torch.fx``; if it is elementwise plus a sum over
    the coordinates (+ - * / neg exp log log1p expm1 pow sigmoid logsigmoid softplus tanh sqrt square abs sin cos, constants
    broadcast along the coordinate axis) its per-coordinate term and the hand-differentiated derivative are emitted as HIP
    source and compiled by ``CTarget.from_source`` (``.traced_source``, ``.compiled``, ``.compiled_form``): the model is then a
    compiled target like a built-in one.  If it is instead a HIERARCHICAL density -- head coordinates ``Th[:, i]``, the rows
    ``Th[:, H:]``, row expressions that may use head-derived values broadcast with ``[:, None]``, sums over ``dim=1``, any scalar
    expression of heads and sums -- it is compiled into the lane-spread form (``trace_lanes.py``) and gets the one-launch HMC
    trajectory / delayed-rejection proposal kernels ``bk.Funnel`` has.  If its coordinates are COUPLED THROUGH SHIFTED SLICES
    (``x[:, 1:] - phi[:, None] * x[:, :-1]``, ``torch.diff``: AR(1) and random-walk priors, state-space and stochastic-volatility
    models) it is compiled into the per-chain form (``trace_chain.py``): compiled gradient op, one launch per leapfrog step and,
    for D <= 128, one launch per trajectory.  Anything else warns (naming the node, ``.compile_note``) and keeps autograd.
    """

    batched = True

    def __init__(self, fn, dims: int, layout: str = "cd", grad_fn=None, compile: bool = False, contract: bool = False):
        if layout not in ("cd", "dc"):
            raise ValueError("layout must be 'cd' (fn takes (C, D), the reference's shape) or 'dc' (fn takes (D, C))")
        self._fn = fn
        self._grad_fn = grad_fn
        self._D = int(dims)
        self._dc = layout == "dc"
        self.compiled = None        # the CTarget the traced source was compiled into (compile=True and traceable)
        self.traced_source = None   # ... and its bk_term source
        self.compile_note = None    # why compile=True kept autograd, if it did
        if grad_fn is not None:
            # (the engine's gradient-only call; absent for autograd, which needs the forward pass anyway.  A namespaced
            # opt-in: a user model's own `gradient` member is never called)
            self.bk_gradient = self._gradient
        if compile:
            self._compile(contract)

    def _compile(self, contract):
        """compile=True: read ``fn`` once with torch.fx; if it is elementwise-plus-sum-over-the-coordinates, emit its
        ``bk_term`` (value and hand-differentiated derivative, trace.py) and run through CTarget.from_source -- the
        model then IS a compiled target for every sampler (streaming gradient kernel, counted launches, the
        whole-draw HMC kernel).  Otherwise: a warning naming the unsupported node, and autograd as before."""
        import warnings

        from . import trace

        form, head = "elementwise", 0
        try:
            src, params, info = trace.term_source(self._fn, self._D, "dc" if self._dc else "cd")
        except trace.Unsupported as e:
            # not separable: a head-plus-sums (hierarchical) density?  (trace_lanes.py; the reference's (C, D) layout only)
            try:
                if self._dc:
                    raise trace.Unsupported("head-plus-sums densities are traced in the (C, D) layout only")
                from . import trace_lanes

                src, head, params, info = trace_lanes.lanes_source(self._fn, self._D)
                form = "lanes"
            except trace.Unsupported as e2:
                # coordinates coupled through shifted slices (AR(1), random walks, state-space models)?  -> the per-chain form
                try:
                    if self._dc:
                        raise trace.Unsupported("traced in the (C, D) layout only")
                    from . import trace_chain

                    src, params, info = trace_chain.chain_source(self._fn, self._D)
                    form = "chain"
                except trace.Unsupported as e3:
                    self.compile_note = (f"as a sum over coordinates: {e}; as head coordinates plus sums over rows: {e2}; "
                                         f"as sums over shifted slices: {e3}")
                    warnings.warn(f"TorchModel(compile=True): not traceable ({self.compile_note}); keeping autograd", stacklevel=3)
                    return
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        p = None if params is None else params.to(dev)
        stage = info.get("stage", "auto") if form == "chain" else "auto"
        try:
            target = CTarget.from_source(src, self._D, params=p, form=form, head=head, contract=contract, stage=stage)
        except _lib.BkHipError as e:
            # no hipcc on this box, a compiler error on the generated text, a refused cache directory, ...: the promise is
            # "anything else warns and keeps autograd", for these as well
            self.compile_note = f"traced as form={form!r}, but the source could not be built: {e}"
            self.traced_source = src
            warnings.warn(f"TorchModel(compile=True): {self.compile_note}; keeping autograd", stacklevel=3)
            return
        if dev.type == "cuda":
            note = self._check_compiled(target, dev)
            if note is not None and note.startswith("not verified"):
                self.compile_note = note    # (accepted: the function could not be evaluated here; say so)
            elif note is not None:
                self.compile_note = note
                warnings.warn(f"TorchModel(compile=True): {note}; keeping autograd", stacklevel=3)
                return
        self.compiled, self.traced_source, self.trace_info = target, src, info
        # the hooks the samplers look for, straight to the compiled target
        self.bk_eval, self.bk_counted = target.bk_eval, target.bk_counted
        self.compiled_form = form
        for name in ("bk_hmc_draw", "bk_hmc_trajectory", "bk_leapfrog_step", "bk_leapfrog_trajectory", "bk_hmc_proposal",
                     "bk_dr_proposal", "bk_dr_proposal_supported", "bk_mala_step"):
            if hasattr(target, name):
                setattr(self, name, getattr(target, name))
        self.__dict__.pop("bk_gradient", None)

    def _check_compiled(self, target, dev):
        """The compiled source against the function itself on a few random points (a tracer records ONE path through
        Python code: a data-dependent branch would be frozen silently).  None if they agree."""
        try:
            g = torch.Generator(device="cpu").manual_seed(20250)
            Theta = (0.5 * torch.randn((16, self._D), generator=g, dtype=torch.float64)).to(dev)
            x = self._arg(Theta).detach().requires_grad_(True)
            with torch.enable_grad():
                lp = self._fn(x)
                (gr,) = torch.autograd.grad(lp.sum(), x)
            gr = gr.t() if self._dc else gr
        except Exception as e:  # the function cannot be evaluated here (e.g. its constants live elsewhere): nothing to compare
            return f"not verified against the function on random points (evaluating it on {dev} raised {type(e).__name__}: {e})"
        lp_c, g_c = target.log_density_gradient(Theta)
        ok = (torch.allclose(lp_c, lp.detach(), rtol=1e-9, atol=1e-9, equal_nan=True)
              and torch.allclose(g_c, gr, rtol=1e-9, atol=1e-9, equal_nan=True))
        return None if ok else "the compiled source disagrees with the function on random points"

    def dims(self) -> int:
        return self._D

    def _arg(self, Theta):
        # Theta: the (C, D) argument of the Model protocol; the samplers pass a transposed view of their (D, C) array
        return Theta.t() if self._dc else Theta

    def log_density(self, Theta):
        if self.compiled is not None:
            return self.compiled.log_density(Theta)
        with torch.no_grad():
            return self._fn(self._arg(Theta))

    def log_density_gradient(self, Theta):
        if self.compiled is not None:
            return self.compiled.log_density_gradient(Theta)
        if self._grad_fn is not None:
            with torch.no_grad():
                x = self._arg(Theta)
                g = self._grad_fn(x)
                return self._fn(x), (g.t() if self._dc else g)
        x = self._arg(Theta).detach().requires_grad_(True)
        with torch.enable_grad():
            lp = self._fn(x)
            (g,) = torch.autograd.grad(lp.sum(), x)
        return lp.detach(), (g.t() if self._dc else g)

    def _gradient(self, Theta):
        """The gradient alone, (C, D) like log_density_gradient's second output (only with grad_fn)."""
        with torch.no_grad():
            g = self._grad_fn(self._arg(Theta))
        return g.t() if self._dc else g
