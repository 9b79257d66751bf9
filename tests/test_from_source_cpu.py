"""CTarget.from_source on the build box (no GPU): hipcc cross-compiles the generated translation units, the libraries
export what the samplers bind, and the cache refuses anything another user could have written (VERDICT r4
item 6).  No compute calls."""
import ctypes
import os
import stat

import pytest

import bayes_kit_amd as bk
from bayes_kit_amd import targets as T

TERM = """
__device__ void bk_term(double th, i64 d, const double* lam, double& term, double& grad) {
  const double t = lam[d] * th; term = -0.5 * (th * t); grad = -t;
}
"""
CHAIN = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double*) {
  double s = 0.0;
  for (i64 d = 0; d < D; ++d) { s = s + th[d] * th[d]; g.set(d, -th[d]); }
  return -0.5 * s;
}
"""
LANES = """
template <class L> __device__ double bk_lanes_density(L& c, const double*) {
  const double v = c.head(0);
  const double s = c.sum([](double x, i64) { return x * x; });
  const double ev = bk_exp(-v);
  c.grad_head(0, -v / 9.0 + 0.5 * ev * s);
  c.grad([ev](double x, i64) { return -(ev * x); });
  return -(v * v) / 18.0 - 0.5 * ev * s;
}
"""


@pytest.fixture()
def cache(tmp_path, monkeypatch):
    d = tmp_path / "cache"
    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(d))
    return d


def exports(lib):
    h = ctypes.CDLL(lib)
    return {n for n in ("bk_src_target", "bk_src_target_n", "bk_src_hmc_draw", "bk_src_hmc_trajectory",
                        "bk_src_dr_proposal_job", "bk_src_leapfrog_step") if hasattr(h, n)}


def test_every_form_compiles_and_exports_its_entry_points(cache):
    e = T._compile_source_target(TERM, "elementwise", False, 16, 0)
    assert exports(e) == {"bk_src_target", "bk_src_target_n", "bk_src_hmc_draw", "bk_src_hmc_trajectory", "bk_src_leapfrog_step",
                          "bk_src_dr_proposal_job"}
    c = T._compile_source_target(CHAIN, "chain", False, 16, 0)
    assert exports(c) == {"bk_src_target", "bk_src_target_n", "bk_src_leapfrog_step"}  # (D <= 128: the one-launch step)
    assert exports(T._compile_source_target(CHAIN, "chain", False, 200, 0)) == {"bk_src_target", "bk_src_target_n"}
    l = T._compile_source_target(LANES, "lanes", False, 101, 1)
    assert exports(l) == {"bk_src_target", "bk_src_target_n", "bk_src_dr_proposal_job", "bk_src_leapfrog_step"}
    big = T._compile_source_target(LANES, "lanes", False, 300, 1)  # > 128 spread rows: gradient op + one-launch step
    assert exports(big) == {"bk_src_target", "bk_src_target_n", "bk_src_leapfrog_step"}
    # the cache: 0700 directory, private files, a second request is served from it (same path, no temporaries left)
    assert stat.S_IMODE(os.lstat(cache).st_mode) == 0o700
    assert T._compile_source_target(TERM, "elementwise", False, 16, 0) == e
    assert sorted(os.listdir(cache)) == sorted([os.path.basename(p) for p in (e, c, l, big)]
                                                + [os.path.basename(T._compile_source_target(CHAIN, "chain", False, 200, 0))])
    for p in (e, c, l, big):
        assert not os.lstat(p).st_mode & 0o022
    # the objects: hooks the samplers look for exist exactly where the library exports them
    te = bk.CTarget.from_source(TERM, 16)
    tl = bk.CTarget.from_source(LANES, 101, form="lanes", head=1)
    tb = bk.CTarget.from_source(LANES, 300, form="lanes", head=1)
    tc = bk.CTarget.from_source(CHAIN, 16, form="chain")
    assert hasattr(te, "bk_hmc_draw") and hasattr(te, "bk_hmc_trajectory") and hasattr(te, "bk_dr_proposal")
    assert hasattr(te, "bk_leapfrog_step") and te.bk_dr_proposal_supported()
    assert hasattr(tl, "bk_dr_proposal") and tl.bk_dr_proposal_supported() and not hasattr(tl, "bk_hmc_draw")
    assert not hasattr(tb, "bk_dr_proposal") and not hasattr(tc, "bk_dr_proposal") and not hasattr(tc, "bk_hmc_draw")
    assert hasattr(tl, "bk_leapfrog_step") and hasattr(tb, "bk_leapfrog_step") and hasattr(tc, "bk_leapfrog_step")
    assert all(t.bk_counted for t in (te, tl, tb, tc))


def test_bad_arguments(cache):
    with pytest.raises(ValueError):
        bk.CTarget.from_source(TERM, 4, form="rows")
    with pytest.raises(ValueError):
        bk.CTarget.from_source(LANES, 4, form="lanes", head=9)
    with pytest.raises(ValueError):
        bk.CTarget.from_source(LANES, 1, form="lanes", head=2)
    with pytest.raises(bk._lib.BkHipError, match="hipcc failed"):
        bk.CTarget.from_source("this is not C++", 3)
    assert os.listdir(cache) == []  # nothing published, no temporaries left behind


def test_cache_directory_others_can_write_is_refused(cache):
    os.makedirs(cache, mode=0o700)
    os.chmod(cache, 0o777)
    with pytest.raises(bk._lib.BkHipError, match="writable"):
        bk.CTarget.from_source(TERM, 4)
    os.chmod(cache, 0o770)
    with pytest.raises(bk._lib.BkHipError, match="writable"):
        bk.CTarget.from_source(TERM, 4)


def test_cache_directory_that_is_a_symlink_is_refused(cache, tmp_path):
    real = tmp_path / "elsewhere"
    real.mkdir(mode=0o700)
    os.symlink(real, cache)
    with pytest.raises(bk._lib.BkHipError, match="symlink"):
        bk.CTarget.from_source(TERM, 4)


def test_cached_library_others_can_write_or_own_is_refused(cache):
    lib = T._compile_source_target(TERM, "elementwise", False, 4, 0)
    os.chmod(lib, 0o666)
    with pytest.raises(bk._lib.BkHipError, match="writable"):
        bk.CTarget.from_source(TERM, 4)
    os.chmod(lib, 0o700)
    bk.CTarget.from_source(TERM, 4)
    # a planted symlink in place of the library
    os.rename(lib, lib + ".real")
    os.symlink(lib + ".real", lib)
    with pytest.raises(bk._lib.BkHipError, match="symlink"):
        bk.CTarget.from_source(TERM, 4)
    os.unlink(lib)
    os.rename(lib + ".real", lib)
    if os.getuid() == 0:  # only root can hand a file to another user
        os.chown(lib, 12345, -1)
        with pytest.raises(bk._lib.BkHipError, match="owned by uid 12345"):
            bk.CTarget.from_source(TERM, 4)
        os.chown(lib, 0, -1)
        os.chown(cache, 12345, -1)
        with pytest.raises(bk._lib.BkHipError, match="owned by uid 12345"):
            bk.CTarget.from_source(TERM, 4)
        os.chown(cache, 0, -1)


def test_default_cache_is_private_and_per_user(tmp_path, monkeypatch):
    monkeypatch.delenv("BK_SOURCE_TARGET_DIR", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "xdg"))
    d = T._source_cache_dir()
    assert d == str(tmp_path / "xdg" / "bayes_kit_amd") and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    monkeypatch.delenv("XDG_CACHE_HOME")
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    assert T._source_cache_dir() == str(tmp_path / "home" / ".cache" / "bayes_kit_amd")
    # no usable home: a fresh mkdtemp directory (0700, unpredictable name), never a fixed path under /tmp
    ro = tmp_path / "ro"
    ro.mkdir()
    (ro / "file").write_text("")
    monkeypatch.setenv("XDG_CACHE_HOME", str(ro / "file"))  # makedirs under a FILE fails for every uid, root included
    monkeypatch.setattr(T, "_PROCESS_CACHE_DIR", None)
    d = T._source_cache_dir()
    assert os.path.basename(d).startswith("bayes_kit_amd_src_") and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    assert T._source_cache_dir() == d
    os.rmdir(d)


def test_missing_hipcc_says_so(cache, monkeypatch):
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("ROCM_PATH", "/nonexistent")
    monkeypatch.setenv("PATH", "")
    with pytest.raises(bk._lib.BkHipError, match="no hipcc was found"):
        bk.CTarget.from_source(TERM + "// never compiled before\n", 4)


def test_stage_option_is_checked(cache):
    for kw, needle in ((dict(form="elementwise", stage="lds"), "form='chain' only"), (dict(form="chain", stage="shared"), "stage must be"),
                       (dict(form="chain", stage="registers", dims=200), "dims <= 128"), (dict(form="chain", stage="lds", dims=400), "dims <= 300")):
        dims = kw.pop("dims", 16)
        with pytest.raises(ValueError, match=needle):
            T._source_text(CHAIN if kw["form"] == "chain" else TERM, kw["form"], dims, 0, kw["stage"])
    text = T._source_text(CHAIN, "chain", 64, 0, "lds")
    assert "#define BK_SOURCE_THETA_LDS 1" in text and "#define BK_SOURCE_STAGE 0" in text and "#define BK_SOURCE_LDS 64" in text
    assert "BK_SOURCE_THETA_LDS" not in T._source_text(CHAIN, "chain", 64, 0)


def test_generated_libraries_match_the_header_of_their_abi(cache):
    """include/bkhip_source.h declares the C ABI of a generated library; every form exports a subset of exactly those names, and
    every declared name is exported by some form."""
    import re

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    src = open(os.path.join(root, "include", "bkhip_source.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\bint\s+(bk_src_[a-z0-9_]+)\s*\(", src))
    assert len(declared) == 9

    def all_exports(lib):
        with open(lib, "rb") as f:
            blob = f.read()
        return set(m.decode() for m in re.findall(rb"(?<=\x00)(bk_src_[a-z0-9_]+)(?=\x00)", blob))

    seen = set()
    for text, form, dims, head, stage in ((TERM, "elementwise", 16, 0, "auto"), (CHAIN, "chain", 16, 0, "auto"),
                                          (CHAIN, "chain", 200, 0, "auto"), (CHAIN, "chain", 16, 0, "lds"),
                                          (LANES, "lanes", 101, 1, "auto"), (LANES, "lanes", 300, 1, "auto")):
        ex = all_exports(T._compile_source_target(text, form, False, dims, head, stage))
        if form == "chain" and dims <= 128:
            assert {"bk_src_leapfrog_step", "bk_src_trajectory"} <= ex, (stage, ex)
        assert ex <= declared and {"bk_src_target", "bk_src_target_n"} <= ex, (form, dims, ex - declared)
        seen |= ex
    assert seen == declared
    # the header compiles as C
    import subprocess

    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-x", "c", "-I", os.path.join(root, "include"),
                           os.path.join(root, "include", "bkhip_source.h")])


def test_prewarm_builds_a_list_of_sources_in_parallel_and_records_what_was_asked_for(cache, tmp_path, monkeypatch):
    """targets.prewarm_sources: a list of from_source specs (as BK_SOURCE_RECORD logs them) built concurrently; duplicates once;
    a source that does not compile is counted, not raised; afterwards the libraries are cache hits."""
    import json
    import time

    rec = tmp_path / "asked.jsonl"
    monkeypatch.setenv("BK_SOURCE_RECORD", str(rec))
    specs = [dict(user_source=TERM, form="elementwise", contract=False, dims=12, head=0, stage="auto"),
             dict(user_source=CHAIN, form="chain", contract=False, dims=12, head=0, stage="auto"),
             dict(user_source=TERM, form="elementwise", contract=False, dims=12, head=0, stage="auto"),   # duplicate
             dict(user_source="__device__ void bk_term(double th) { this does not compile }", form="elementwise",
                  contract=False, dims=12, head=0, stage="auto")]
    errors = []
    ok, bad = T.prewarm_sources(specs, workers=3, errors=errors)
    assert (ok, bad) == (2, 1) and len(errors) == 1 and errors[0][:2] == ("elementwise", 12)
    asked = [json.loads(line) for line in open(rec)]
    assert len(asked) == 3 and {a["form"] for a in asked} == {"elementwise", "chain"}   # (the duplicate was not asked for twice)
    # two dims of a source whose generated text does not depend on D are ONE library built by two threads at once: both must
    # succeed (per-thread temporary names; the GPU suite's manifest holds such pairs)
    twins = [dict(user_source=TERM.replace("th", "th "), form="elementwise", contract=False, dims=d, head=0, stage="auto") for d in (20, 21)]
    assert T.prewarm_sources(twins, workers=2) == (2, 0)
    t0 = time.perf_counter()
    lib = T._compile_source_target(TERM, "elementwise", False, 12, 0)
    assert os.path.exists(lib) and time.perf_counter() - t0 < 2.0     # a cache hit
    # the committed manifest of the GPU test-suite parses and has the fields prewarm_sources reads
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    manifest = [json.loads(line) for line in open(os.path.join(root, "tests", "golden", "from_source_manifest.jsonl"))]
    assert len(manifest) > 40 and all({"user_source", "form", "dims"} <= set(m) for m in manifest)


def test_cache_below_a_directory_others_can_write_is_refused(tmp_path, monkeypatch):
    """a leaf that is private is not enough -- a group / world-writable directory ABOVE the cache (without the sticky
    bit) lets somebody else swap the cache directory itself.  A sticky one (/tmp-like) is fine: the leaf is ours."""
    loose = tmp_path / "shared"
    loose.mkdir()
    os.chmod(loose, 0o777)
    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(loose / "cache"))
    with pytest.raises(bk._lib.BkHipError, match="above the cache directory"):
        T._source_cache_dir()
    os.chmod(loose, 0o1777)   # sticky, like /tmp
    assert T._source_cache_dir() == str(loose / "cache")
    os.chmod(loose, 0o755)
    assert T._source_cache_dir() == str(loose / "cache")


def test_default_cache_below_a_loose_directory_falls_back_to_a_private_temporary_one(tmp_path, monkeypatch):
    """The DEFAULT location ($XDG_CACHE_HOME) below a world-writable directory: not trusted, but not fatal either."""
    loose = tmp_path / "xdg"
    loose.mkdir()
    os.chmod(loose, 0o777)
    monkeypatch.delenv("BK_SOURCE_TARGET_DIR", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(loose))
    monkeypatch.setattr(T, "_PROCESS_CACHE_DIR", None)
    with pytest.warns(UserWarning, match="private temporary directory"):
        d = T._source_cache_dir()
    assert not d.startswith(str(loose)) and stat.S_IMODE(os.stat(d).st_mode) == 0o700
    os.chmod(loose, 0o755)
