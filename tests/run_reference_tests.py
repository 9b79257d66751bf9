#!/usr/bin/env python3
"""Run the REFERENCE'S OWN test-suite against bayes_kit_amd (build container only).

    python tests/run_reference_tests.py [pytest args]

`bayes_kit` and its sub-modules are aliased to `bayes_kit_amd` before pytest collects
/root/reference/test (read in place: nothing is copied, nothing is written there), so every
`from bayes_kit... import ...` in those tests resolves to the drop-in.  There is no GPU in
the build container, so the device operations are served by tests/fake_ops.py (the CPU
stand-in used for host-logic tests): this checks the API surface, argument validation, call
contracts, seeding and moment behaviour of the Python layer -- not the kernels, which the
`-m gpu` tests cover.  Last result: 69 passed (all of the reference's tests).  The unseeded
moment tests are statistical: test_drghmc_binom fails about 15 % of runs with the reference
itself (8/50) and with this package (7/50); with equal seeds the two produce bit-identical
draws (800 of 800 checked), so a rare failure there is the test's own noise; likewise
test_iat_ar1 (unseeded AR(1) data) fails about 1 run in 6 with the reference and with this package.
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]

import bayes_kit_amd as bk  # noqa: E402
from tests.fake_ops import FakeOps  # noqa: E402

bk._lib._default_ops = FakeOps()
sys.modules["bayes_kit"] = bk
for name in ("hmc", "mala", "drghmc", "metropolis", "rhat", "ess", "iat", "autocorr", "smc", "typing", "ensemble"):
    sys.modules["bayes_kit." + name] = importlib.import_module("bayes_kit_amd." + name)

if not os.path.isdir(os.path.join(REF, "test")):
    sys.exit("the reference is only mounted in the build container")
os.chdir(REF)
sys.path.insert(0, REF)
import pytest  # noqa: E402

sys.exit(pytest.main(["test", "-p", "no:cacheprovider", "-q"] + sys.argv[1:]))
