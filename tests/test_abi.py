"""The C-ABI library loads and exports every symbol include/bkhip.h declares (no GPU)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bkhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|double|int64_t)\s+(bk_[a-z0-9_]+)\s*\(", src)))


def test_build_and_symbols():
    import __graft_entry__ as ge

    ge.build()
    from bayes_kit_amd import _lib

    lib = ctypes.CDLL(_lib.lib_path())
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bkhip.h but not exported"
    # and the binding covers exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.load().bk_version() >= 100


def test_product_rng_source_on_host_matches_numpy():
    # bk_host_* run the SAME source as the kernels (csrc/bk_rng.hpp), compiled for the host
    from bayes_kit_amd import _lib

    lib = _lib.load()
    key = [77, 5]
    st = np.zeros(_lib.RNG_WORDS, dtype=np.uint64)
    st[0], st[1], st[10] = key[0], key[1], 4
    n = 400_000
    out = np.empty(n)
    assert lib.bk_host_normals(_lib.RNG_PHILOX, st.ctypes.data, out.ctypes.data, n) == 0
    g = np.random.Generator(np.random.Philox(key=key))
    assert np.array_equal(g.normal(size=n).view(np.uint64), out.view(np.uint64))
    u = np.empty(9)
    assert lib.bk_host_uniforms(_lib.RNG_PHILOX, st.ctypes.data, u.ctypes.data, 9) == 0
    assert np.array_equal(u, g.uniform(size=9))
    s = g.bit_generator.state
    assert [int(v) for v in s["state"]["counter"]] == [int(v) for v in st[2:6]]
    assert [int(v) for v in s["buffer"]] == [int(v) for v in st[6:10]] and s["buffer_pos"] == int(st[10])
    # PCG64 = np.random.default_rng(int)
    g = np.random.default_rng(2024)
    from bayes_kit_amd._engine import _bitgen_words

    kind, w = _bitgen_words(g.bit_generator)
    assert kind == _lib.RNG_PCG64
    out = np.empty(100_000)
    assert lib.bk_host_normals(kind, w.ctypes.data, out.ctypes.data, len(out)) == 0
    assert np.array_equal(g.normal(size=len(out)).view(np.uint64), out.view(np.uint64))
    import math

    for x in -np.random.default_rng(0).random(20000):
        assert lib.bk_host_log1p(float(x)) == math.log1p(float(x))
    # bk_exp (include/bkhip_math.h: fma-based, a specified operation sequence): the library's host build == the oracle's
    # restatement (exact rational fma), bit for bit, over tiny / moderate / large arguments, subnormal results, both clamps
    # and the specials; and within one ulp of the host libm
    from oracle.rng import exp_bk

    g = np.random.default_rng(3)
    xs = np.concatenate([g.normal(size=20000) * 4.0, g.uniform(-0.4, 0.4, 5000), g.uniform(-1.1, 1.1, 5000), g.uniform(-745.2, 709.8, 5000),
                         g.normal(size=2000) * 1e-9, [0.0, -0.0, 1e-300, -1e-300, 709.78, 709.79, -745.13, -745.14, -708.4, -740.0,
                                                       float("inf"), float("-inf")]])
    for x in xs:
        a, b = lib.bk_host_exp(float(x)), exp_bk(float(x))
        assert a == b, (x, a, b)
        if np.isfinite(a) and a > 1e-300:
            assert abs(a - math.exp(float(x))) <= abs(np.nextafter(a, np.inf) - a), x
    assert math.isnan(lib.bk_host_exp(float("nan"))) and math.isnan(exp_bk(float("nan")))


def test_example_plugin_target_loads_and_exports_its_entry_point():
    """The plugin ABI (bk_target_fn in include/bkhip.h): a user library exporting one function."""
    import ctypes
    import os

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    assert "typedef int (*bk_target_fn)(" in open(os.path.join(root, "include", "bkhip.h")).read()
    lib = ctypes.CDLL(os.path.join(root, "examples", "plugin_target", "libar1_target.so"))
    assert hasattr(lib, "ar1_target")
    # argument validation happens before any launch: callable without a GPU
    lib.ar1_target.restype = ctypes.c_int
    lib.ar1_target.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_int64, ctypes.c_void_p]
    assert lib.ar1_target(None, None, None, 0, None, 1, 1, None) == -1
    # a target with the counted form as well (bk_target_fn_n: chain count from device memory)
    assert "typedef int (*bk_target_fn_n)(" in open(os.path.join(root, "include", "bkhip.h")).read()
    lib2 = ctypes.CDLL(os.path.join(root, "examples", "plugin_target", "libfunnel_target.so"))
    for name, extra in (("funnel_target", []), ("funnel_target_n", [ctypes.c_void_p])):
        fn = getattr(lib2, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + extra + [
            ctypes.c_void_p]
        assert fn(*([None, None, None, 0, None, 1, 1] + [None] * len(extra) + [None])) == -1


def test_error_behaviour_of_the_c_abi_without_a_gpu():
    """Status codes of include/bkhip.h: NULL / bad extents -> BK_E_ARG (-1), ld < C -> BK_E_ALIGN (-2),
    empty problems (C == 0) -> BK_OK with nothing launched.  All three are decided before any HIP
    call, so they can be checked on a machine without a GPU (dummy non-NULL pointers are never read)."""
    import ctypes

    from bayes_kit_amd import _lib

    lib = _lib.load()
    buf = (ctypes.c_double * 64)()
    p = ctypes.addressof(buf)
    OK, E_ARG, E_ALIGN = 0, -1, -2
    # empty problems
    assert lib.bk_momentum_refresh(0, p, 0, None, 0.0, 1.0, p, 0, None, None, None, 0, 8, None, 0, None) == OK
    assert lib.bk_log_uniform(0, p, 0, p, None, 0, None) == OK
    assert lib.bk_leapfrog_kick_drift(p, p, p, p, 0, p, 0, 1, None, 0.1, 0, 0.0, 1, 0.1, 0, 8, None) == OK
    assert lib.bk_leapfrog_kick_drift(p, p, p, p, 4, p, 4, 1, None, 0.1, 0, 0.0, 1, 0.1, 4, 0, None) == OK  # D = 0
    assert lib.bk_leapfrog_finish(p, None, 0, p, 0, 1, None, 0.1, 0, p, 0, 8, None) == OK
    assert lib.bk_mh_accept(0, p, p, p, p, p, p, p, None, 0, None) == OK
    assert lib.bk_select_columns(p, p, p, None, None, None, 0, 0, 8, None) == OK
    assert lib.bk_mala_logq(p, p, p, p, 0, 0.1, p, p, 0, 8, None) == OK
    assert lib.bk_target_diag_gaussian_grad(p, p, None, 0, p, 0, 8, None) == OK
    assert lib.bk_welford_update(p, p, 0, p, 0, 1, 0, 8, None) == OK
    assert lib.bk_ess(p, 0, 8, 0, p, None, 0, None) == OK
    assert lib.bk_end_pos_pairs(p, 0, 0, p, 0, None) == OK
    # argument errors
    assert lib.bk_momentum_refresh(0, None, 4, None, 0.0, 1.0, p, 4, None, None, None, 4, 8, None, 0, None) == E_ARG
    assert lib.bk_momentum_refresh(7, p, 4, None, 0.0, 1.0, p, 4, None, None, None, 4, 8, None, 0, None) == E_ARG  # rng kind
    assert lib.bk_leapfrog_kick_drift(None, p, p, p, 4, p, 4, 1, None, 0.1, 0, 0.0, 1, 0.1, 4, 8, None) == E_ARG
    assert lib.bk_leapfrog_kick_drift(p, p, p, p, 4, p, 4, 1, None, 0.1, 0, 0.0, 1, 0.1, -1, 8, None) == E_ARG
    assert lib.bk_mh_accept(5, p, p, p, p, p, p, p, None, 4, None) == E_ARG      # unknown accept mode
    assert lib.bk_select_columns(p, p, p, p, None, None, 4, 4, 8, None) == E_ARG  # dst1 without src1
    assert lib.bk_target_diag_gaussian_grad(p, p, None, 4, None, 4, 8, None) == E_ARG  # lam required
    assert lib.bk_ess(p, 4, 3, 0, p, None, 4, None) == E_ARG                     # N < 4 (ess.py:67-68)
    assert lib.bk_autocorr(p, 4, 1, p, 4, 4, None) == E_ARG                      # N < 2 (autocorr.py:23-24)
    assert lib.bk_normals_chain_major(1, p, 4, p, 8, 4, 8, None, 0, None) == E_ARG        # Philox streams only
    assert lib.bk_normals_chain_major(0, p, 4, p, 7, 4, 8, None, 0, None) == E_ARG        # ldz < D
    # round-3 entry points
    assert lib.bk_autocorr_fft(p, 4, 1, p, 4, 4, p, 1 << 20, None) == E_ARG         # N < 2
    assert lib.bk_autocorr_fft(p, 4, 8, p, 4, 4, p, 16, None) == E_ARG              # scratch too small
    assert lib.bk_autocorr_fft(p, 4, 8, p, 4, 0, p, 16, None) == OK                 # no columns
    assert lib.bk_autocorr_fft_work_bytes(8, 4) > 2 * 16 * 2 * 16 and lib.bk_autocorr_fft_work_bytes(1, 4) == 0
    assert lib.bk_sort_by_key(p, p, p, p, 4, p, 1 << 20, None) == E_ARG             # in and out must differ
    assert lib.bk_sort_by_key_work_bytes(1 << 31) == -1 and lib.bk_sort_by_key_work_bytes(0) == 0
    assert lib.bk_dr_refresh_begin(0, p, 4, p, 0.9, 0.4, p, 4, None, None, 4, 8, None, 0, p, p, p, p, p, 1.0, None, 0,
                                   None, None, None) == E_ARG                        # kinetic energy output required
    assert lib.bk_dr_refresh_begin(0, p, 4, p, 0.9, 0.4, p, 4, None, p, 0, 8, None, 0, p, p, p, p, p, 1.0, None, 0,
                                   None, None, None) == OK                           # no chains
    job = _lib.DiagJob(None, 4, 4, 8, p, p, p, 4, 0, None, None, 0, None, 0, 0)     # a side job without its input
    assert lib.bk_dr_refresh_begin(0, p, 4, p, 0.9, 0.4, p, 4, None, p, 0, 8, None, 0, p, p, p, p, p, 1.0, None, 0,
                                   None, ctypes.addressof(job), None) == E_ARG
    g0 = _lib.Ghost0(0.1, 0, None, 1.0, None, None, None, None)                      # a ghost of zero steps
    assert lib.bk_dr_proposal_funnel(p, p, p, 4, None, p, p, p, p, p, 4, None, 0.1, 3, 4, 8, None, None, None, p, p, p,
                                         None, None, ctypes.byref(g0), None) == E_ARG
    # layout errors
    assert lib.bk_leapfrog_kick_drift(p, p, p, p, 3, p, 4, 1, None, 0.1, 0, 0.0, 1, 0.1, 4, 8, None) == E_ALIGN
    assert lib.bk_select_columns(p, p, p, None, None, None, 3, 4, 8, None) == E_ALIGN
    assert lib.bk_mala_logq(p, p, p, p, 3, 0.1, p, p, 4, 8, None) == E_ALIGN
    assert lib.bk_mala_propose_from_normals(p, p, p, 5, 3, p, 4, 0.1, 0.2, 4, 8, None) == E_ALIGN  # z strides
    assert lib.bk_refresh_work_elems(5, 33) == 5 * 40


def test_fft_plan_of_the_reference_call_shape_is_not_padded_to_a_wide_batch():
    """ess(chain) / autocorr(chain) on ONE chain (the reference's call shape, ess.py:52-69): the plan's scratch is
    two complex arrays of the transform size, not 36 columns of them (10 GB for 4M draws)."""
    from bayes_kit_amd import _lib

    lib = _lib.load()
    for N, C in ((1_000_000, 1), (4_000_000, 2), (100_000, 7)):
        size = 1 << (2 * N - 2).bit_length()
        assert size >= 2 * N - 1
        cp = (C + 1) // 2
        assert lib.bk_autocorr_fft_work_bytes(N, C) <= 2 * size * cp * 16 + size * 16 + (1 << 20), (N, C)
    # wide batches keep their rows off the power-of-two pitch
    assert lib.bk_autocorr_fft_work_bytes(16384, 4096) > 2 * 32768 * 2048 * 16


def test_source_targets_compile_without_a_gpu_and_export_both_plugin_forms(tmp_path, monkeypatch):
    """CTarget.from_source: the generated translation unit cross-compiles with hipcc (no GPU needed) and exports the
    host-sized and the counted plugin entry points; argument validation happens before any launch."""
    import ctypes

    from bayes_kit_amd import targets

    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(tmp_path))
    elem = "__device__ void bk_term(double th, i64 d, const double* p, double& term, double& grad) { term = -0.5 * th * th; grad = -th; }"
    chain = ("__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* p) {"
             " double s = 0.0; for (i64 d = 0; d < D; ++d) { s = s + th[d] * th[d]; g.set(d, -th[d]); } return -0.5 * s; }")
    for form, src in (("elementwise", elem), ("chain", chain)):
        path = targets._compile_source_target(src, form, False)
        assert path.startswith(str(tmp_path)) and targets._compile_source_target(src, form, False) == path  # cached by content
        lib = ctypes.CDLL(path)
        for name, extra in (("bk_src_target", 0), ("bk_src_target_n", 1)):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + [
                ctypes.c_void_p] * (1 + extra)
            assert fn(*([None, None, None, 0, None, 1, 1] + [None] * (1 + extra))) == -1
    import pytest

    with pytest.raises(ValueError):
        targets._compile_source_target(elem, "rowwise", False)
    with pytest.raises(targets._lib.BkHipError):
        targets._compile_source_target("not C++", "elementwise", False)
