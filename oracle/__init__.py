"""CPU oracle for the many-chain HMC / MALA / DRGHMC hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker (or as the thing timed as the CPU baseline), never as
a compute path of ``bayes_kit_amd``.

It is a restatement, in this repo's own NumPy code, of the algorithms in the reference
(flatironinstitute/bayes-kit, cited per function as ``bayes_kit/<file>:<lines>``).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (build container only), runs it on seeded inputs and commits the
outputs under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
oracle function against those vectors and against the literal known answers held by the
reference's own tests (test_rhat.py, test_autocorr.py, test_iat.py).
"""
