"""Module-path parity with ``bayes_kit/autocorr.py``."""
from .diagnostics import autocorr  # noqa: F401
