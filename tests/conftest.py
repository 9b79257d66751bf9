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
