"""Module-path parity with ``bayes_kit/rhat.py``."""
from .diagnostics import (rank_chains, rank_normalize_chains, rank_normalized_rhat, rhat, split_chains,  # noqa: F401
                          split_rhat)
