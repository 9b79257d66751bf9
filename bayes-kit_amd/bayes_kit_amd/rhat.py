"""Module-path parity with ``bayes_kit/rhat.py``."""
from .diagnostics import rhat, split_chains, split_rhat  # noqa: F401
