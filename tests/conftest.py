import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "bayes-kit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the in-tree libbkhip.so normally travels with the working tree; if it does not, build it
    # (hipcc cross-compiles without a GPU).  A failed build surfaces in the tests that load it.
    lib = os.path.join(ROOT, "bayes-kit_amd", "bayes_kit_amd", "lib", "libbkhip.so")
    plugin = os.path.join(ROOT, "examples", "plugin_target", "libar1_target.so")
    plugin2 = os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so")
    c_host = os.path.join(ROOT, "examples", "c_host", "hmc_main")
    if not all(os.path.exists(f) for f in (lib, plugin, plugin2, c_host)):
        try:
            import __graft_entry__ as ge

            ge.build()
        except Exception as e:  # pragma: no cover
            print("could not build libbkhip.so:", e)
    _prewarm_from_source_cache(session)


def _usable_cpus():
    # the CPUs this process may actually run on at once: affinity capped by the cgroup quota (256 visible, 16 usable on the
    # pool's boxes: 256 hipcc processes at once would only thrash)
    import math

    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return n


def _prewarm_from_source_cache(session):
    """`pytest -m gpu` on a fresh box compiles ~57 from_source densities with hipcc, one after the other: 184 of the suite's
    236 seconds (the same suite with a warm cache: 52 s).  tests/golden/from_source_manifest.jsonl is the log of what the
    suite asked for (BK_SOURCE_RECORD); build all of it first, in parallel.  A stale manifest only costs time."""
    import json

    m = session.config.getoption("-m") or ""
    if "gpu" not in m or "not gpu" in m or os.environ.get("BK_TEST_NO_PREWARM"):
        return
    path = os.path.join(ROOT, "tests", "golden", "from_source_manifest.jsonl")
    try:
        import torch

        if not torch.cuda.is_available() or not os.path.exists(path):
            return
        from bayes_kit_amd import targets

        specs = [json.loads(line) for line in open(path) if line.strip()]
        import time

        t0 = time.perf_counter()
        errors = []
        ok, bad = targets.prewarm_sources(specs, workers=_usable_cpus(), errors=errors)
        print(f"[conftest] from_source cache: {ok} libraries ready, {bad} failed, {time.perf_counter() - t0:.0f} s")
        for form, dims, msg in errors:  # (the suite holds sources that must NOT compile; anything else shows here)
            print(f"[conftest]   failed: form={form} D={dims}: ...{msg[-160:]!r}")
    except Exception as e:  # pragma: no cover  (the tests compile what they need themselves)
        print("[conftest] from_source prewarm skipped:", e)


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible and they were not
    # explicitly selected with -m gpu; with -m gpu on a box without a GPU they fail loudly.
    import torch

    if torch.cuda.is_available():
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
