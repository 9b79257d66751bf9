"""ctypes binding of libbkhip.so (the C ABI declared in include/bkhip.h).

There is NO fallback: if the shared library is missing, or a sampler is asked to run
without a visible MI355X, the error is raised to the caller.  ``import torch`` happens
before the library is loaded so that libbkhip.so binds to the HIP runtime PyTorch already
brought into the process (same SONAME), which is what makes PyTorch's stream handles and
device pointers valid inside the library.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_int, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  (must precede the CDLL below)

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libbkhip.so")

RNG_WORDS = 11
RNG_PHILOX = 0
RNG_PCG64 = 1
ACCEPT_HMC = 0
ACCEPT_MALA = 1

P = c_void_p
I = c_int64
F = c_double

# name -> argument types (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    "bk_version": [],
    "bk_rng_init_philox": [P, I, c_uint64, c_uint64, I, P],
    "bk_refresh_work_elems": [I, I],
    "bk_momentum_refresh": [c_int, P, I, P, F, F, P, I, P, P, P, I, I, P, I, P],
    "bk_log_uniform": [c_int, P, I, P, P, I, P],
    "bk_uniform": [c_int, P, I, P, P, I, P],
    "bk_leapfrog_kick_drift": [P, P, P, P, I, P, I, I, P, F, c_int, F, c_int, F, I, I, P],
    "bk_leapfrog_finish": [P, P, I, P, I, I, P, F, c_int, P, I, I, P],
    "bk_leapfrog_kick_drift_n": [P, P, P, P, I, P, I, I, P, F, c_int, F, c_int, F, I, I, P, P],
    "bk_leapfrog_first_step_gather": [P, P, P, I, P, P, P, I, P, F, F, I, I, P, P],
    "bk_leapfrog_finish_level": [P, P, I, P, I, I, P, F, c_int, P, I, I, P, P, P, P, P, P, P, P],
    "bk_mh_accept": [c_int, P, P, P, P, P, P, P, P, I, P],
    "bk_select_columns": [P, P, P, P, P, P, I, I, I, P],
    "bk_blend_columns": [P, P, P, P, I, I, I, P],
    "bk_compact_indices": [P, I, P, P, P, P],
    "bk_dr_begin": [P, P, P, P, P, P, I, P],
    "bk_dr_retry_test": [c_int, P, I, P, F, P, I, P],
    "bk_dr_level_begin": [P, P, P, P, P, I, P, P],
    "bk_dr_ghost_update": [P, P, I, P, P, P, P, P],
    "bk_dr_accept_prob": [P, P, P, P, P, F, P, P, I, P, P],
    "bk_dr_accept_test": [c_int, P, I, P, P, P, I, P, P, P, P, P, P, P],
    "bk_dr_accept_prob_test": [c_int, P, I, P, P, P, P, P, F, I, P, P, P, P, P, P, P],
    "bk_dr_accept_prob_ghost": [P, P, P, P, P, F, P, P, I, P, P, P, P],
    "bk_dr_begin_retry": [c_int, P, I, P, P, P, P, P, P, F, P, I, P, I, P],
    "bk_dr_refresh_begin": [c_int, P, I, P, F, F, P, I, P, P, I, I, P, I, P, P, P, P, P, F, P, I, P, P, P],
    "bk_dr_accept_prob_test_next": [c_int, P, I, P, P, P, P, P, F, I, P, P, P, P, P, P, P, P, P],
    "bk_dr_accept_prob_ghost_next": [P, P, P, P, P, F, P, P, I, P, P, P, P, P, P],
    "bk_scatter_columns": [P, P, I, I, P, P, P, P, P, P, I, I, P, P, P, P],
    "bk_mala_propose": [c_int, P, I, P, P, P, I, F, F, I, I, P],
    "bk_mala_propose_from_normals": [P, P, P, I, I, P, I, F, F, I, I, P],
    "bk_normals_chain_major": [c_int, P, I, P, I, I, I, P, I, P],
    "bk_mala_logq": [P, P, P, P, I, F, P, P, I, I, P],
    "bk_mala_single_draw": [c_int, P, I, P, P, P, P, P, P, P, I, P, P, F, F, I, c_int, F, P],
    "bk_mala_step_supported": [I, I, I],
    "bk_mala_step": [P, P, P, P, P, I, P, P, P, P, I, F, F, P, P, P, I, I, P],
    "bk_mala_step_gaussian": [P, P, P, I, P, P, P, P, P, I, F, F, P, P, P, I, I, P],
    "bk_target_iso_gaussian_grad": [P, P, P, I, I, I, P],
    "bk_target_diag_gaussian_grad": [P, P, P, I, P, I, I, P],
    "bk_target_funnel_grad": [P, P, P, I, I, I, P],
    "bk_target_iso_gaussian_grad_n": [P, P, P, I, I, I, P, P],
    "bk_target_diag_gaussian_grad_n": [P, P, P, I, P, I, I, P, P],
    "bk_target_funnel_grad_n": [P, P, P, I, I, I, P, P],
    "bk_leapfrog_step_funnel": [P, P, I, P, F, I, I, P, P],
    "bk_leapfrog_step_gaussian": [P, P, I, P, P, F, I, I, P, P],
    "bk_dr_proposal_gaussian": [P, P, P, I, P, P, P, P, P, P, I, P, F, I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    "bk_hmc_trajectory_funnel": [P, P, P, P, P, P, P, I, P, F, I, I, I, P],
    "bk_hmc_trajectory_gaussian": [P, P, P, P, I, P, P, F, I, I, I, P],
    "bk_hmc_draw_gaussian": [P, P, I, P, P, I, P, P, F, I, P, P, P, P, P, P, P, P, P, I, I, P],
    "bk_dr_proposal_funnel": [P, P, P, I, P, P, P, P, P, P, I, P, F, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "bk_dense_metric_apply": [P, I, P, P, I, I, I, P],
    "bk_gemm_chains": [P, I, I, I, P, I, P, I, I, P, I, P],
    "bk_gemm_chains_work_elems": [I, I, I],
    "bk_logistic_residual": [P, I, P, P, I, I, I, P],
    "bk_logistic_finish": [P, P, I, P, I, F, F, P, P, P, I, I, P],
    "bk_dot_columns": [P, P, I, F, P, I, I, P],
    "bk_resample_indices": [P, I, P, I, P, P, P],
    "bk_gather_columns": [P, P, I, P, I, I, I, P],
    "bk_relayout": [P, I, I, P, I, I, I, I, P],
    "bk_welford_update": [P, P, I, P, I, I, I, I, P],
    "bk_record_series": [P, I, P, I, P, P, I, I, I, P],
    "bk_record_series_dev": [P, I, P, I, P, P, I, P, I, I, P],
    "bk_welford_update_dev": [P, P, I, P, I, P, I, I, I, P],
    "bk_rhat_partials": [P, P, I, I, P, P, I, I, P],
    "bk_chain_mean_var": [P, I, P, I, P, P, I, P],
    "bk_end_pos_pairs": [P, I, I, P, I, P],
    "bk_ess": [P, I, I, c_int, P, P, I, P],
    "bk_iat_from_acor": [P, I, I, c_int, P, P, I, P],
    "bk_gemm_chains_logistic": [P, I, I, I, P, I, P, I, I, P, P],
    "bk_autocorr": [P, I, I, P, I, I, P],
    "bk_autocorr_fft_work_bytes": [I, I],
    "bk_autocorr_fft": [P, I, I, P, I, I, P, I, P],
    "bk_rank_normalize": [P, F, P, I, P],
    "bk_sort_by_key_work_bytes": [I],
    "bk_sort_by_key": [P, P, P, P, I, P, I, P],
    "bk_count_below": [P, I, P, I, P, P],
    "bk_scatter_ranks": [P, I, F, P, P],
    "bk_host_normals": [c_int, P, P, I],
    "bk_host_uniforms": [c_int, P, P, I],
    "bk_host_log1p": [F],
    "bk_host_exp": [F],
}
_RESTYPE = {"bk_gemm_chains_work_elems": c_int64, "bk_host_log1p": c_double, "bk_host_exp": c_double, "bk_refresh_work_elems": c_int64, "bk_sort_by_key_work_bytes": c_int64,
             "bk_autocorr_fft_work_bytes": c_int64}


class BkHipError(RuntimeError):
    pass


class GhostLink(ctypes.Structure):
    """bk_ghost_link of include/bkhip.h: what a ghost proposal without ghosts of its own owes its parent level."""
    _fields_ = [("parent_H", P), ("parent_h", P), ("parent_live", P), ("parent_a", P), ("a_out", P),
                ("prob_retry", F), ("next_index", P), ("next_count", P)]


class Ghost0(ctypes.Structure):
    """bk_ghost0 of include/bkhip.h: the first ghost of the proposals a launch produces, run by that launch."""
    _fields_ = [("h", F), ("steps", I), ("parent_a", P), ("prob_retry", F), ("next_index", P), ("next_count", P),
                ("lanes_out", P), ("lanes_total", P)]


class DiagJob(ctypes.Structure):
    """bk_diag_job of include/bkhip.h: the previous draw's welford_update_dev and / or record_series_dev call, carried along
    by a draw's generator launch."""
    _fields_ = [("theta", P), ("ld_theta", I), ("C", I), ("D", I), ("n_dev", P), ("mean", P), ("m2", P), ("ld", I),
                ("n_offset", I), ("series", P), ("dims", P), ("K", I), ("logp", P), ("capacity", I), ("row_offset", I)]


class ScatterJob(ctypes.Structure):
    """bk_scatter_job of include/bkhip.h: the arguments of one bk_scatter_columns call."""
    _fields_ = [("mask", P), ("index", P), ("n", I), ("D", I), ("dst0", P), ("src0", P), ("dst1", P), ("src1", P),
                ("dst2", P), ("src2", P), ("ld_dst", I), ("ld_src", I), ("sdst", P), ("ssrc", P), ("n_dev", P)]


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load libbkhip.so (once) and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise BkHipError(
            f"{_LIB_PATH} is missing: build it with `python bayes-kit_amd/build.py` "
            "(or __graft_entry__.build()).  There is no CPU fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def require_gpu() -> torch.device:
    """The sampler path needs a real device; fail loudly otherwise."""
    load()
    if not torch.cuda.is_available():
        raise BkHipError(
            "bayes_kit_amd needs an AMD Instinct GPU (gfx950) visible to PyTorch-ROCm; "
            "no device found and there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())


class RawDeviceArray:
    """A float64 array in device memory of its own (hipMalloc / hipFree through the HIP runtime PyTorch loaded), outside
    PyTorch's caching allocator: for scratch whose release must not touch the user's allocator state (no
    torch.cuda.empty_cache()).  ``tensor()`` is a PyTorch view of it (via __cuda_array_interface__) that keeps this
    object alive; the memory is freed when the last reference goes."""

    _hip = None
    live = 0  # arrays currently allocated (tests)

    def __init__(self, shape):
        if RawDeviceArray._hip is None:
            RawDeviceArray._hip = ctypes.CDLL("libamdhip64.so")
        self.shape = tuple(int(x) for x in shape)
        n = 8
        for x in self.shape:
            n *= x
        p = c_void_p()
        rc = RawDeviceArray._hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(max(n, 8)))
        if rc != 0 or not p.value:
            raise BkHipError(f"hipMalloc of {n} bytes failed: hipError_t {rc}")
        self._ptr = p.value
        RawDeviceArray.live += 1
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f8", "data": (self._ptr, False), "version": 2,
                                         "strides": None}

    def tensor(self):
        t = torch.as_tensor(self, device=torch.device("cuda", torch.cuda.current_device()))
        t._bk_raw = self  # (belt and braces: the view holds its owner)
        return t

    def __del__(self):
        p, self._ptr = getattr(self, "_ptr", None), None
        cls = type(self)  # (at interpreter shutdown the module's globals -- the class name included -- are already None)
        hip = getattr(cls, "_hip", None)
        if p and hip is not None:
            cls.live -= 1
            try:
                hip.hipFree(ctypes.c_void_p(p))  # (hipFree waits for work in flight on the device)
            except Exception:  # interpreter shutdown
                pass


def check(status: int, what: str) -> None:
    if status != 0:
        kind = "argument/layout error" if status < 0 else "hipError_t"
        raise BkHipError(f"{what} failed: {kind} {status}")


def ptr(t) -> int:
    """Raw device (or host) pointer of a tensor, 0 for None."""
    return 0 if t is None else t.data_ptr()


def stream_handle() -> int:
    """hipStream_t of PyTorch's current stream (all kernels are enqueued on it)."""
    return torch.cuda.current_stream().cuda_stream


def _ld(t) -> int:
    """Leading dimension of a [D, n] chain-contiguous tensor."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise BkHipError(f"expected a [D, C] tensor with contiguous chains, got strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class Ops:
    """Tensor-level call layer over the C ABI (one method per entry point).

    Samplers talk to the device only through an ``Ops`` object.  The one shipped here is
    the HIP library and nothing else; tests may inject an object with the same methods to
    exercise the host-side control flow of a sampler on a machine without a GPU.

    Tensor conventions: phase-space arrays are fp64 ``[D, n]`` tensors whose chain axis is
    contiguous (``stride == (ld, 1)``); per-chain vectors are ``[n]``; RNG tables are
    uint64-as-int64 ``[RNG_WORDS, C]``; masks are uint8 ``[n]``.
    """

    name = "hip"

    def __init__(self):
        self.lib = load()
        self.device = require_gpu()
        # optional per-launch HIP-event timing: {entry point: [(start, end), ...]} on the
        # stream the kernels are enqueued on (bench.py's roofline measurement).  timed_stride = n: only every n-th
        # launch of a timed entry point is bracketed by events -- an event record is a packet of its own on the
        # queue (about 6 us each: 256 of them cost a config-3 draw 1.6 ms = 4 % when every launch was bracketed)
        self.timed = None
        self.timed_stride = 1
        self._timed_seen = {}

    # -- helpers ---------------------------------------------------------------------
    def _call(self, name, *args):
        rec = self.timed.get(name) if self.timed is not None else None
        if rec is not None and self.timed_stride > 1:
            k = self._timed_seen.get(name, 0)
            self._timed_seen[name] = k + 1
            if k % self.timed_stride:
                rec = None
        if rec is None:
            check(getattr(self.lib, name)(*args), name)
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        check(getattr(self.lib, name)(*args), name)
        e1.record()
        rec.append((e0, e1))

    @staticmethod
    def _s():
        return stream_handle()

    # -- RNG ---------------------------------------------------------------------------
    def rng_init_philox(self, state, key0, chain_id0):
        self._call("bk_rng_init_philox", ptr(state), state.stride(0), key0 & (2**64 - 1),
                   chain_id0 & (2**64 - 1), state.shape[1], self._s())

    def refresh_work(self, C, D):
        """Scratch for the wavefront-per-chain momentum refresh (bk_refresh_work_elems doubles)."""
        return torch.empty(self.lib.bk_refresh_work_elems(C, D), dtype=torch.float64, device=self.device)

    def momentum_refresh(self, kind, state, loc_in, loc_mul, scale, out, metric, kin_out, active=None, work=None):
        D, C = out.shape
        self._call("bk_momentum_refresh", kind, ptr(state), state.stride(0), ptr(loc_in), loc_mul, scale,
                   ptr(out), _ld(out), ptr(metric), ptr(kin_out), ptr(active), C, D, ptr(work),
                   0 if work is None else work.numel(), self._s())

    def log_uniform(self, kind, state, out, active=None):
        self._call("bk_log_uniform", kind, ptr(state), state.stride(0), ptr(out), ptr(active),
                   out.shape[0], self._s())

    def uniform(self, kind, state, out, active=None):
        self._call("bk_uniform", kind, ptr(state), state.stride(0), ptr(out), ptr(active), out.shape[0],
                   self._s())

    # -- integrator -----------------------------------------------------------------------
    # n_dev (optional; here and in target_grad): int32 device tensor [1] with the number of lanes really in the
    # set -- the tensors' chain extent is then only the bound the launch is sized for (bk_*_n entry points)
    def kick_drift(self, theta_in, theta_out, rho_in, rho_out, grad, metric, eps,
                   use_pre, pre, use_kick, kick, n_dev=None):
        D, C = theta_out.shape
        ld = _ld(theta_out)
        assert _ld(theta_in) == ld and _ld(rho_in) == ld and _ld(rho_out) == ld
        if n_dev is None:
            self._call("bk_leapfrog_kick_drift", ptr(theta_in), ptr(theta_out), ptr(rho_in), ptr(rho_out), ld,
                       ptr(grad), grad.stride(0), grad.stride(1), ptr(metric), eps, int(use_pre), pre,
                       int(use_kick), kick, C, D, self._s())
        else:
            self._call("bk_leapfrog_kick_drift_n", ptr(theta_in), ptr(theta_out), ptr(rho_in), ptr(rho_out), ld,
                       ptr(grad), grad.stride(0), grad.stride(1), ptr(metric), eps, int(use_pre), pre,
                       int(use_kick), kick, C, D, ptr(n_dev), self._s())

    def first_step_gather(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, metric, eps, pre,
                          n_dev=None):
        D, n = theta_out.shape
        ld_in = _ld(theta_in)
        assert _ld(rho_in) == ld_in and _ld(grad_in) == ld_in and _ld(rho_out) == _ld(theta_out)
        self._call("bk_leapfrog_first_step_gather", ptr(theta_in), ptr(rho_in), ptr(grad_in), ld_in,
                   ptr(src_index), ptr(theta_out), ptr(rho_out), _ld(theta_out), ptr(metric), eps, pre,
                   n, D, ptr(n_dev), self._s())

    def leapfrog_finish(self, rho_in, rho_out, grad, metric, half, negate, kin_out, n_dev=None, level=None,
                        lanes_out=None, lanes_total=None):
        """level: optional (logp, H, h, live) -- the dr_level_begin of the lanes the trajectory produced, done by
        the same launch; lanes_out / lanes_total: the launch's lane count (written / added)."""
        D, C = rho_in.shape
        if rho_out is not None:
            assert _ld(rho_out) == _ld(rho_in)
        gs = (0, 0) if grad is None else grad.stride()
        if n_dev is None and level is None and lanes_out is None and lanes_total is None:
            self._call("bk_leapfrog_finish", ptr(rho_in), ptr(rho_out), _ld(rho_in), ptr(grad), gs[0], gs[1],
                       ptr(metric), half, int(negate), ptr(kin_out), C, D, self._s())
            return
        logp, H, h, live = level if level is not None else (None,) * 4
        self._call("bk_leapfrog_finish_level", ptr(rho_in), ptr(rho_out), _ld(rho_in), ptr(grad), gs[0], gs[1],
                   ptr(metric), half, int(negate), ptr(kin_out), C, D, ptr(n_dev), ptr(logp), ptr(H), ptr(h),
                   ptr(live), ptr(lanes_out), ptr(lanes_total), self._s())

    def mh_accept(self, mode, lp_cur, a_cur, lp_prop, a_prop, log_u, mask, ret, count):
        self._call("bk_mh_accept", mode, ptr(lp_cur), ptr(a_cur), ptr(lp_prop), ptr(a_prop), ptr(log_u),
                   ptr(mask), ptr(ret), ptr(count), lp_cur.shape[0], self._s())

    def select_columns(self, mask, dst0, src0, dst1=None, src1=None, copy0=None):
        D, C = dst0.shape
        ld = _ld(dst0)
        assert _ld(src0) == ld and (dst1 is None or (_ld(dst1) == ld and _ld(src1) == ld))
        assert copy0 is None or _ld(copy0) == ld
        self._call("bk_select_columns", ptr(mask), ptr(dst0), ptr(src0), ptr(dst1), ptr(src1), ptr(copy0), ld,
                   C, D, self._s())

    # -- delayed rejection ---------------------------------------------------------------------
    # n_dev (optional, everywhere below): int32 device tensor [1] holding the number of lanes really
    # in the set; `n` / `m` is then only the bound the launch is sized for (include/bkhip.h).

    def blend_columns(self, mask, a, b, out):
        """out = mask ? b : a (a, b read-only): the select of a sampler that rebinds its state array."""
        D, C = a.shape
        ld = _ld(a)
        assert _ld(b) == ld and _ld(out) == ld
        self._call("bk_blend_columns", ptr(mask), ptr(a), ptr(b), ptr(out), ld, C, D, self._s())

    def compact_indices(self, mask, n, idx_out, count_out, n_dev=None):
        self._call("bk_compact_indices", ptr(mask), n, ptr(idx_out), ptr(count_out), ptr(n_dev), self._s())

    def dr_begin(self, logp, kin, cur_H, cur_h, rej, alive):
        self._call("bk_dr_begin", ptr(logp), ptr(kin), ptr(cur_H), ptr(cur_h), ptr(rej), ptr(alive),
                   logp.shape[0], self._s())

    def dr_retry_test(self, kind, state, rej, prob_retry, alive):
        self._call("bk_dr_retry_test", kind, ptr(state), state.stride(0), ptr(rej), float(prob_retry),
                   ptr(alive), alive.shape[0], self._s())

    def dr_level_begin(self, logp, kin, H, h, live, n, n_dev=None):
        self._call("bk_dr_level_begin", ptr(logp), ptr(kin), ptr(H), ptr(h), ptr(live), n, ptr(n_dev), self._s())

    def dr_ghost_update(self, ga, sub_index, m, h, live, a, n_dev=None):
        self._call("bk_dr_ghost_update", ptr(ga), ptr(sub_index), m, ptr(h), ptr(live), ptr(a), ptr(n_dev),
                   self._s())

    def dr_accept_prob(self, H, cur_H, h, cur_h, cur_index, prob_retry, live, a, n, n_dev=None):
        self._call("bk_dr_accept_prob", ptr(H), ptr(cur_H), ptr(h), ptr(cur_h), ptr(cur_index),
                   float(prob_retry), ptr(live), ptr(a), n, ptr(n_dev), self._s())

    def dr_accept_test(self, kind, state, chain_index, a, H, n, cur_H, cur_h, rej, alive, accepted, n_dev=None):
        self._call("bk_dr_accept_test", kind, ptr(state), state.stride(0), ptr(chain_index), ptr(a), ptr(H),
                   n, ptr(cur_H), ptr(cur_h), ptr(rej), ptr(alive), ptr(accepted), ptr(n_dev), self._s())

    def dr_accept_prob_test(self, kind, state, chain_index, H, h, live, a, prob_retry, n, cur_H, cur_h, rej, alive,
                            accepted, n_dev=None):
        """dr_accept_prob against the chains' current point + dr_accept_test, one launch."""
        self._call("bk_dr_accept_prob_test", kind, ptr(state), state.stride(0), ptr(chain_index), ptr(H), ptr(h),
                   ptr(live), ptr(a), float(prob_retry), n, ptr(cur_H), ptr(cur_h), ptr(rej), ptr(alive),
                   ptr(accepted), ptr(n_dev), self._s())

    def dr_accept_prob_ghost(self, H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, parent_live, parent_a,
                             n_dev=None):
        """dr_accept_prob of a ghost level + dr_ghost_update of its parent level, one launch."""
        self._call("bk_dr_accept_prob_ghost", ptr(H), ptr(parent_H), ptr(h), ptr(parent_h), ptr(sub_index),
                   float(prob_retry), ptr(live), ptr(a), n, ptr(n_dev), ptr(parent_live), ptr(parent_a), self._s())

    def dr_begin_retry(self, kind, state, logp, kin, cur_H, cur_h, rej, alive, prob_retry, counters, draw_counter=None):
        """dr_begin + the first stage's dr_retry_test + zeroing of the draw's lane counters (+ the sampler's
        device-side draw counter incremented), one launch."""
        self._call("bk_dr_begin_retry", kind, ptr(state), state.stride(0), ptr(logp), ptr(kin), ptr(cur_H), ptr(cur_h),
                   ptr(rej), ptr(alive), float(prob_retry), ptr(counters), 0 if counters is None else counters.numel(),
                   ptr(draw_counter), logp.shape[0], self._s())

    def dr_refresh_begin(self, kind, state, loc_in, loc_mul, scale, out, metric, kin_out, work, logp, cur_H, cur_h, rej,
                         alive, prob_retry, counters, draw_counter=None, side=None):
        """momentum_refresh(..., kin_out) + dr_begin_retry(...): the generator's launch + one more.  side: a diag_job(...)
        done by workgroups of the generator's launch (the previous draw's update of attached moments / tracked series)."""
        D, C = out.shape
        self._call("bk_dr_refresh_begin", kind, ptr(state), state.stride(0), ptr(loc_in), loc_mul, scale, ptr(out),
                   _ld(out), ptr(metric), ptr(kin_out), C, D, ptr(work), 0 if work is None else work.numel(), ptr(logp),
                   ptr(cur_H), ptr(cur_h), ptr(rej), ptr(alive), float(prob_retry), ptr(counters),
                   0 if counters is None else counters.numel(), ptr(draw_counter),
                   None if side is None else ctypes.cast(ctypes.pointer(side), P), self._s())

    def diag_job(self, theta, n_dev, welford=None, record=None):
        """welford_update_dev(mean, m2, theta, n_dev, n_offset) and / or record_series_dev(theta, dims, logp, series, n_dev,
        row_offset) as ONE job for dr_refresh_begin(side=...).  welford = (mean, m2, n_offset); record = (dims, logp, series,
        row_offset)."""
        D, C = theta.shape
        assert n_dev.dtype == torch.int64 and (welford is not None or record is not None)
        j = DiagJob(ptr(theta), _ld(theta), C, D, ptr(n_dev), None, None, 0, 0, None, None, 0, None, 0, 0)
        if welford is not None:
            mean, m2, n_offset = welford
            assert _ld(m2) == _ld(mean)
            j.mean, j.m2, j.ld, j.n_offset = ptr(mean), ptr(m2), _ld(mean), int(n_offset)
        if record is not None:
            dims, logp, series, row_offset = record
            j.series, j.dims, j.K, j.logp = ptr(series), ptr(dims), (0 if dims is None else dims.numel()), ptr(logp) or None
            j.capacity, j.row_offset = series.shape[1], int(row_offset)
        return j

    def dr_accept_prob_test_next(self, kind, state, chain_index, H, h, live, a, prob_retry, n, cur_H, cur_h, rej, alive,
                                 accepted, next_index, next_count, n_dev=None):
        """dr_accept_prob_test + the next stage's retry test + the list of chains that propose again."""
        self._call("bk_dr_accept_prob_test_next", kind, ptr(state), state.stride(0), ptr(chain_index), ptr(H), ptr(h),
                   ptr(live), ptr(a), float(prob_retry), n, ptr(cur_H), ptr(cur_h), ptr(rej), ptr(alive),
                   ptr(accepted), ptr(n_dev), ptr(next_index), ptr(next_count), self._s())

    def dr_accept_prob_ghost_next(self, H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, parent_live,
                                  parent_a, next_index, next_count, n_dev=None):
        """dr_accept_prob_ghost + the list of parent lanes that go on to their next ghost."""
        self._call("bk_dr_accept_prob_ghost_next", ptr(H), ptr(parent_H), ptr(h), ptr(parent_h), ptr(sub_index),
                   float(prob_retry), ptr(live), ptr(a), n, ptr(n_dev), ptr(parent_live), ptr(parent_a),
                   ptr(next_index), ptr(next_count), self._s())

    def scatter_columns(self, mask, index, n, dsts, srcs, sdst=None, ssrc=None, n_dev=None):
        """dsts/srcs: up to three [D, *] tensors each (same ld within each list)."""
        d = list(dsts) + [None] * (3 - len(dsts))
        s_ = list(srcs) + [None] * (3 - len(srcs))
        D = dsts[0].shape[0]
        self._call("bk_scatter_columns", ptr(mask), ptr(index), n, D, ptr(d[0]), ptr(s_[0]), ptr(d[1]),
                   ptr(s_[1]), ptr(d[2]), ptr(s_[2]), _ld(dsts[0]), _ld(srcs[0]), ptr(sdst), ptr(ssrc),
                   ptr(n_dev), self._s())

    # -- MALA --------------------------------------------------------------------------------
    def mala_propose(self, kind, state, theta, grad, theta_prop, eps, sqrt2eps):
        D, C = theta.shape
        ld = _ld(theta)
        assert _ld(grad) == ld and _ld(theta_prop) == ld
        self._call("bk_mala_propose", kind, ptr(state), state.stride(0), ptr(theta), ptr(grad),
                   ptr(theta_prop), ld, eps, sqrt2eps, C, D, self._s())

    def mala_propose_from_normals(self, theta, grad, z, theta_prop, eps, sqrt2eps):
        """z: logical [D, C]; either laid out like theta or the transposed view of chain-major
        normals (``zt[:, :D].t()`` of a normals_chain_major buffer)."""
        D, C = theta.shape
        ld = _ld(theta)
        assert _ld(grad) == ld and _ld(theta_prop) == ld and tuple(z.shape) == (D, C)
        self._call("bk_mala_propose_from_normals", ptr(theta), ptr(grad), ptr(z), z.stride(0), z.stride(1),
                   ptr(theta_prop), ld, eps, sqrt2eps, C, D, self._s())

    def normals_chain_major(self, kind, state, zt, D, snapshot=None, max_workgroups=0):
        """zt[c, :D] = the next D standard normals of chain c; `snapshot` (optional, a table like
        `state`) receives the stream table as it was before the call; max_workgroups > 0: a background launch
        of at most that many workgroups (bk_normals_chain_major)."""
        C = zt.shape[0]
        assert zt.stride(1) == 1 and zt.shape[1] >= D
        assert snapshot is None or (snapshot.shape == state.shape and snapshot.stride(0) == state.stride(0))
        self._call("bk_normals_chain_major", kind, ptr(state), state.stride(0), ptr(zt), zt.stride(0), C, D,
                   ptr(snapshot), int(max_workgroups or 0), self._s())

    def mala_logq(self, theta, grad, theta_prop, grad_prop, eps, lp_forward, lp_reverse):
        D, C = theta.shape
        ld = _ld(theta)
        assert _ld(grad) == ld and _ld(theta_prop) == ld and _ld(grad_prop) == ld
        self._call("bk_mala_logq", ptr(theta), ptr(grad), ptr(theta_prop), ptr(grad_prop), ld, eps,
                   ptr(lp_forward), ptr(lp_reverse), C, D, self._s())

    def mala_step_supported(self, C, D, ld):
        return bool(self.lib.bk_mala_step_supported(C, D, ld))

    def mala_step(self, theta, theta_out, grad, theta_prop, grad_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps,
                  mask, ret, count):
        """Proposal densities + accept + select (+ the next proposal from the chain-major normals
        zt_next[c, :D], or None) in one pass; theta_out may be theta itself."""
        D, C = theta.shape
        ld = _ld(theta)
        assert _ld(theta_out) == ld and _ld(grad) == ld and _ld(theta_prop) == ld and _ld(grad_prop) == ld
        ldz = 0
        if zt_next is not None:
            assert zt_next.shape[0] == C and zt_next.stride(1) == 1 and zt_next.shape[1] >= D
            ldz = zt_next.stride(0)
        self._call("bk_mala_step", ptr(theta), ptr(theta_out), ptr(grad), ptr(theta_prop), ptr(grad_prop), ld,
                   ptr(lp), ptr(lp_prop), ptr(log_u), ptr(zt_next), ldz, eps, sqrt2eps, ptr(mask), ptr(ret),
                   ptr(count), C, D, self._s())

    def mala_step_gaussian(self, lam, theta, theta_out, theta_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret,
                           count):
        """mala_step for the separable built-in Gaussians (lam None = identity): both gradients recomputed from theta /
        theta_prop inside the kernel, none stored (bk_mala_step_gaussian)."""
        D, C = theta.shape
        ld = _ld(theta)
        assert _ld(theta_out) == ld and _ld(theta_prop) == ld
        ldz = 0
        if zt_next is not None:
            assert zt_next.shape[0] == C and zt_next.stride(1) == 1 and zt_next.shape[1] >= D
            ldz = zt_next.stride(0)
        self._call("bk_mala_step_gaussian", ptr(theta), ptr(theta_out), ptr(theta_prop), ld, ptr(lam), ptr(lp), ptr(lp_prop),
                   ptr(log_u), ptr(zt_next), ldz, eps, sqrt2eps, ptr(mask), ptr(ret), ptr(count), C, D, self._s())

    # -- built-in targets -----------------------------------------------------------------------
    def target_grad(self, kind, params, theta, grad, logp, n_dev=None):
        D, C = theta.shape
        ld = _ld(theta)
        if grad is not None:
            assert _ld(grad) == ld
        if n_dev is not None:
            if kind == "iso_gaussian":
                self._call("bk_target_iso_gaussian_grad_n", ptr(theta), ptr(grad), ptr(logp), ld, C, D, ptr(n_dev),
                           self._s())
            elif kind == "diag_gaussian":
                self._call("bk_target_diag_gaussian_grad_n", ptr(theta), ptr(grad), ptr(logp), ld, ptr(params),
                           C, D, ptr(n_dev), self._s())
            elif kind == "funnel":
                self._call("bk_target_funnel_grad_n", ptr(theta), ptr(grad), ptr(logp), ld, C, D, ptr(n_dev),
                           self._s())
            else:
                raise BkHipError(f"built-in target {kind!r} has no counted gradient")
        elif kind == "iso_gaussian":
            self._call("bk_target_iso_gaussian_grad", ptr(theta), ptr(grad), ptr(logp), ld, C, D, self._s())
        elif kind == "diag_gaussian":
            self._call("bk_target_diag_gaussian_grad", ptr(theta), ptr(grad), ptr(logp), ld, ptr(params),
                       C, D, self._s())
        elif kind == "funnel":
            self._call("bk_target_funnel_grad", ptr(theta), ptr(grad), ptr(logp), ld, C, D, self._s())
        else:
            raise BkHipError(f"unknown built-in target {kind!r}")

    def hmc_trajectory_funnel(self, theta_in, rho, grad_in, theta_out, grad_out, logp_out, kin_out, metric, eps, steps):
        """A whole HMC trajectory on the funnel in ONE launch (rho is overwritten)."""
        D, C = theta_in.shape
        ld = _ld(theta_in)
        assert all(_ld(t) == ld for t in (rho, grad_in, theta_out, grad_out))
        self._call("bk_hmc_trajectory_funnel", ptr(theta_in), ptr(rho), ptr(grad_in), ptr(theta_out), ptr(grad_out),
                   ptr(logp_out), ptr(kin_out), ld, ptr(metric), eps, steps, C, D, self._s())

    def leapfrog_step_funnel(self, theta, rho, metric, h, n_dev=None):
        """One leapfrog step {gradient, kick, drift} on the funnel in ONE launch, theta / rho [D, n] advanced in place."""
        D, n = theta.shape
        ld = _ld(theta)
        assert _ld(rho) == ld
        self._call("bk_leapfrog_step_funnel", ptr(theta), ptr(rho), ld, ptr(metric), h, n, D, ptr(n_dev), self._s())

    def hmc_trajectory_gaussian(self, theta_in, theta_out, rho_in, rho_out, lam, metric, eps, steps):
        D, C = theta_in.shape
        ld = _ld(theta_in)
        assert _ld(theta_out) == ld and _ld(rho_in) == ld and _ld(rho_out) == ld
        self._call("bk_hmc_trajectory_gaussian", ptr(theta_in), ptr(theta_out), ptr(rho_in), ptr(rho_out), ld,
                   ptr(lam), ptr(metric), eps, steps, C, D, self._s())

    def hmc_draw_gaussian(self, theta_in, theta_out, rho_in, zt, lam, metric, eps, steps, part, kin0, kin1, lp_out,
                          accept=None):
        """Trajectory + both kinetic energies + end-point log density of one HMC draw; momentum from
        rho_in ([D, C]) or from chain-major normals zt ([C, >=D]), exactly one of them.
        accept: optional (lp_cur, log_u, mask, ret, count) -- the draw's accept test in the same launches
        (what mh_accept(ACCEPT_HMC, lp_cur, kin0, lp_out, kin1, log_u, mask, ret, count) does)."""
        D, C = theta_in.shape
        ld = _ld(theta_in)
        assert _ld(theta_out) == ld and (rho_in is None or _ld(rho_in) == ld) and part.numel() >= 12 * C
        ldz = 0
        if zt is not None:
            assert zt.shape[0] == C and zt.stride(1) == 1 and zt.shape[1] >= D
            ldz = zt.stride(0)
        lp_cur, log_u, mask, ret, count = accept if accept is not None else (None,) * 5
        self._call("bk_hmc_draw_gaussian", ptr(theta_in), ptr(theta_out), ld, ptr(rho_in), ptr(zt), ldz, ptr(lam),
                   ptr(metric), eps, steps, ptr(part), ptr(kin0), ptr(kin1), ptr(lp_out), ptr(lp_cur), ptr(log_u),
                   ptr(mask), ptr(ret), ptr(count), C, D, self._s())

    def scatter_job(self, mask, index, n, dsts, srcs, sdst=None, ssrc=None, n_dev=None):
        """The arguments of scatter_columns(...) as a job for dr_proposal_funnel(job=...)."""
        d = list(dsts) + [None] * (3 - len(dsts))
        s_ = list(srcs) + [None] * (3 - len(srcs))
        return ScatterJob(ptr(mask) or None, ptr(index) or None, n, dsts[0].shape[0], ptr(d[0]) or None, ptr(s_[0]) or None,
                          ptr(d[1]) or None, ptr(s_[1]) or None, ptr(d[2]) or None, ptr(s_[2]) or None, _ld(dsts[0]),
                          _ld(srcs[0]), ptr(sdst) or None, ptr(ssrc) or None, ptr(n_dev) or None)

    def ghost_link(self, parent_H, parent_h, parent_live, parent_a, a_out, prob_retry, next_index=None, next_count=None):
        """dr_accept_prob_ghost[_next] of a ghost level without ghosts of its own, as a part of its proposal
        launch (dr_proposal_funnel(ghost=...))."""
        return GhostLink(ptr(parent_H), ptr(parent_h), ptr(parent_live), ptr(parent_a), ptr(a_out), float(prob_retry),
                         ptr(next_index) or None, ptr(next_count) or None)

    def ghost0(self, h, steps, parent_a, prob_retry, next_index=None, next_count=None, lanes_out=None, lanes_total=None):
        """The first ghost of the proposals a dr_proposal_funnel launch produces, integrated and applied to the
        produced level by that launch (dr_proposal_funnel(ghost0=...))."""
        return Ghost0(float(h), int(steps), ptr(parent_a), float(prob_retry), ptr(next_index) or None,
                      ptr(next_count) or None, ptr(lanes_out) or None, ptr(lanes_total) or None)

    def dr_proposal_gaussian(self, lam, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                             kin_out, metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None,
                             ghost=None, ghost0=None):
        """dr_proposal_funnel for the separable Gaussians (lam None: isotropic); D <= 128."""
        D, n = theta_out.shape
        H, hh, live = level if level is not None else (None, None, None)
        ld_in = _ld(theta_in)
        assert _ld(rho_in) == ld_in and (grad_in is None or _ld(grad_in) == ld_in)
        ld_out = _ld(theta_out)
        assert _ld(rho_out) == ld_out and (grad_out is None or _ld(grad_out) == ld_out)
        self._call("bk_dr_proposal_gaussian", ptr(theta_in), ptr(rho_in), ptr(grad_in), ld_in, ptr(src_index),
                   ptr(theta_out), ptr(rho_out), ptr(grad_out), ptr(logp_out), ptr(kin_out), ld_out, ptr(metric),
                   h, steps, n, D, ptr(n_dev), ptr(lanes_out), ptr(lanes_total), ptr(H), ptr(hh), ptr(live),
                   None if job is None else ctypes.byref(job), None if ghost is None else ctypes.byref(ghost),
                   None if ghost0 is None else ctypes.byref(ghost0), ptr(lam), self._s())

    def leapfrog_step_gaussian(self, lam, theta, rho, metric, h, n_dev=None):
        """One leapfrog step {gradient, kick, drift} on a separable Gaussian in ONE launch, theta / rho advanced in place."""
        D, n = theta.shape
        ld = _ld(theta)
        assert _ld(rho) == ld
        self._call("bk_leapfrog_step_gaussian", ptr(theta), ptr(rho), ld, ptr(lam), ptr(metric), h, n, D, ptr(n_dev), self._s())

    def dr_proposal_funnel(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                           kin_out, metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None,
                           ghost=None, ghost0=None):
        """level: optional (H, h, live) tensors of the destination level -- its bk_dr_level_begin is then
        done by the same launch.  job: optional scatter_job(...) run by surplus workgroups of the launch.
        ghost: optional ghost_link(...), the level's accept probability + parent update in the same launch.
        ghost0: optional ghost0(...), the produced level's first ghost in the same launch."""
        D, n = theta_out.shape
        H, hh, live = level if level is not None else (None, None, None)
        ld_in = _ld(theta_in)
        assert _ld(rho_in) == ld_in and (grad_in is None or _ld(grad_in) == ld_in)
        ld_out = _ld(theta_out)
        assert _ld(rho_out) == ld_out and (grad_out is None or _ld(grad_out) == ld_out)
        args = (ptr(theta_in), ptr(rho_in), ptr(grad_in), ld_in, ptr(src_index),
                ptr(theta_out), ptr(rho_out), ptr(grad_out), ptr(logp_out), ptr(kin_out), ld_out, ptr(metric),
                h, steps, n, D, ptr(n_dev), ptr(lanes_out), ptr(lanes_total), ptr(H), ptr(hh), ptr(live))
        self._call("bk_dr_proposal_funnel", *args, None if job is None else ctypes.byref(job),
                   None if ghost is None else ctypes.byref(ghost),
                   None if ghost0 is None else ctypes.byref(ghost0), self._s())

    def dense_metric_apply(self, M, X, Y):
        D, C = X.shape
        assert _ld(Y) == _ld(X) and M.stride(1) == 1
        self._call("bk_dense_metric_apply", ptr(M), M.stride(0), ptr(X), ptr(Y), _ld(X), C, D, self._s())

    def gemm_chains_work(self, R, K, C):
        """Split-K scratch for gemm_chains (a tensor of bk_gemm_chains_work_elems doubles, or None: no split)."""
        n = int(self.lib.bk_gemm_chains_work_elems(R, K, C))
        return torch.empty(n, dtype=torch.float64, device=self.device) if n > 0 else None

    def gemm_chains(self, A, X, Y, work=None):
        """Y[R, C] = A[R, K] @ X[K, C] on the fp64 matrix cores (work: optional split-K scratch)."""
        R, K = A.shape
        assert X.shape[0] == K and Y.shape[0] == R and A.stride(1) == 1
        self._call("bk_gemm_chains", ptr(A), A.stride(0), R, K, ptr(X), _ld(X), ptr(Y), _ld(Y), X.shape[1],
                   ptr(work), 0 if work is None else work.numel(), self._s())

    def gemm_chains_logistic(self, A, X, Y, y_rows):
        """Y[R, C] = y_rows[:, None] - sigmoid(A[R, K] @ X[K, C]): gemm_chains + logistic_residual(part=None) in one launch."""
        R, K = A.shape
        assert X.shape[0] == K and Y.shape[0] == R and A.stride(1) == 1 and y_rows.numel() == R
        self._call("bk_gemm_chains_logistic", ptr(A), A.stride(0), R, K, ptr(X), _ld(X), ptr(Y), _ld(Y), X.shape[1],
                   ptr(y_rows), self._s())

    def logistic_residual(self, Z, y, part, segments=None):
        """part None: gradient only (no log likelihood formed); `segments` then still says how many row blocks the
        launch is cut into (its parallelism over the observations)."""
        N, C = Z.shape
        self._call("bk_logistic_residual", ptr(Z), _ld(Z), ptr(y), ptr(part), N, C,
                   part.shape[0] if part is not None else int(segments or 256), self._s())

    def logistic_finish(self, G, theta, part, inv_prior_var, t, grad, logp, loglik):
        D, C = theta.shape
        ld = _ld(theta)
        assert (G is None or _ld(G) == ld) and (grad is None or _ld(grad) == ld)
        self._call("bk_logistic_finish", ptr(G), ptr(theta), ld, ptr(part), 1 if part is None else part.shape[0], inv_prior_var, t,
                   ptr(grad), ptr(logp), ptr(loglik), C, D, self._s())

    def dot_columns(self, x, y, scale, out):
        D, C = x.shape
        assert _ld(y) == _ld(x)
        self._call("bk_dot_columns", ptr(x), ptr(y), _ld(x), scale, ptr(out), C, D, self._s())

    def resample_indices(self, weights, u, cdf_work, idx_out):
        self._call("bk_resample_indices", ptr(weights), weights.shape[0], ptr(u), u.shape[0], ptr(cdf_work),
                   ptr(idx_out), self._s())

    def gather_columns(self, index, src, dst):
        D, m = dst.shape
        self._call("bk_gather_columns", ptr(index), ptr(src), _ld(src), ptr(dst), _ld(dst), m, D, self._s())

    def relayout(self, src, dst):
        """dst[d, c] = src[d, c] for logical [D, C] tensors of any strides (LDS-tiled)."""
        D, C = dst.shape
        self._call("bk_relayout", ptr(src), src.stride(0), src.stride(1), ptr(dst), dst.stride(0),
                   dst.stride(1), C, D, self._s())

    # -- diagnostics --------------------------------------------------------------------------------
    def welford_update(self, mean, m2, theta, n):
        """theta may have a row pitch of its own (padded sampler state); mean and m2 share theirs."""
        D, C = theta.shape
        assert _ld(m2) == _ld(mean)
        self._call("bk_welford_update", ptr(mean), ptr(m2), _ld(mean), ptr(theta), _ld(theta), n, C, D, self._s())

    def record_series(self, theta, dims, logp, series, row):
        """series[k, row, :] = theta[dims[k], :] (k < K), series[K, row, :] = logp: one launch."""
        D, C = theta.shape
        assert series.is_contiguous() and series.shape[2] == C
        self._call("bk_record_series", ptr(theta), _ld(theta), ptr(dims), 0 if dims is None else dims.numel(), ptr(logp),
                   ptr(series), series.shape[1], row, C, self._s())

    def record_series_dev(self, theta, dims, logp, series, row_dev, row_offset):
        """record_series with the row index read on the device: row = row_dev[0] - row_offset."""
        D, C = theta.shape
        assert series.is_contiguous() and series.shape[2] == C and row_dev.dtype == torch.int64
        self._call("bk_record_series_dev", ptr(theta), _ld(theta), ptr(dims), 0 if dims is None else dims.numel(),
                   ptr(logp), ptr(series), series.shape[1], ptr(row_dev), int(row_offset), C, self._s())

    def welford_update_dev(self, mean, m2, theta, n_dev, n_offset):
        """welford_update with n = n_dev[0] - n_offset read on the device; theta may have its own row pitch."""
        D, C = theta.shape
        assert _ld(m2) == _ld(mean) and n_dev.dtype == torch.int64
        self._call("bk_welford_update_dev", ptr(mean), ptr(m2), _ld(mean), ptr(theta), _ld(theta), ptr(n_dev),
                   int(n_offset), C, D, self._s())

    def rhat_partials(self, mean, m2, n, center, out):
        D, C = mean.shape
        assert _ld(m2) == _ld(mean)
        self._call("bk_rhat_partials", ptr(mean), ptr(m2), _ld(mean), n, ptr(center), ptr(out), C, D,
                   self._s())

    def chain_mean_var(self, x, lengths, mean, var):
        N, C = x.shape
        self._call("bk_chain_mean_var", ptr(x), _ld(x), ptr(lengths), N, ptr(mean), ptr(var), C, self._s())

    def rank_normalize(self, rank, S, out):
        self._call("bk_rank_normalize", ptr(rank), float(S), ptr(out), rank.numel(), self._s())

    def sort_by_key(self, keys, vals):
        """Stable ascending sort of (float64 key, int64 payload) pairs -> (sorted keys, payloads)."""
        n = keys.numel()
        ko, vo = torch.empty_like(keys), torch.empty_like(vals)
        if n == 0:
            return ko, vo
        nb = self.lib.bk_sort_by_key_work_bytes(n)
        if nb < 0:
            raise BkHipError("bk_sort_by_key_work_bytes failed")
        work = torch.empty(max(1, nb), dtype=torch.uint8, device=keys.device)
        self._call("bk_sort_by_key", ptr(keys), ptr(ko), ptr(vals), ptr(vo), n, ptr(work), work.numel(), self._s())
        return ko, vo

    def count_below(self, sorted_keys, queries):
        out = torch.empty(queries.numel(), dtype=torch.int64, device=queries.device)
        self._call("bk_count_below", ptr(sorted_keys), sorted_keys.numel(), ptr(queries), queries.numel(), ptr(out),
                   self._s())
        return out

    def scatter_ranks(self, payload, base, out):
        self._call("bk_scatter_ranks", ptr(payload), payload.numel(), float(base), ptr(out), self._s())

    def autocorr(self, x, out):
        N, C = x.shape
        self._call("bk_autocorr", ptr(x), _ld(x), N, ptr(out), _ld(out), C, self._s())

    def autocorr_fft_work_bytes(self, N, C):
        return int(self.lib.bk_autocorr_fft_work_bytes(N, C))

    def autocorr_fft(self, x, out):
        """autocorr(x, out) by FFT (long chains): the library's own Stockham passes, scratch allocated here."""
        N, C = x.shape
        work = torch.empty(max(16, self.lib.bk_autocorr_fft_work_bytes(N, C)), dtype=torch.uint8, device=x.device)
        self._call("bk_autocorr_fft", ptr(x), _ld(x), N, ptr(out), _ld(out), C, ptr(work), work.numel(), self._s())

    def end_pos_pairs(self, acor, out):
        N, C = acor.shape
        self._call("bk_end_pos_pairs", ptr(acor), max(C, acor.stride(0)) if N else C, N, ptr(out), C, self._s())

    def iat_from_acor(self, acor, estimator, ess_out, iat_out=None):
        N, C = acor.shape
        self._call("bk_iat_from_acor", ptr(acor), _ld(acor), N, estimator, ptr(ess_out), ptr(iat_out), C, self._s())

    def ess(self, x, estimator, ess_out, iat_out=None):
        N, C = x.shape
        self._call("bk_ess", ptr(x), _ld(x), N, estimator, ptr(ess_out), ptr(iat_out), C, self._s())


_default_ops = None


def default_ops() -> Ops:
    """The process-wide HIP ops object (raises without the library or without a GPU)."""
    global _default_ops
    if _default_ops is None:
        _default_ops = Ops()
    return _default_ops
