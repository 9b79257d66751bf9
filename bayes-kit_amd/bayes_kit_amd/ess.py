"""Module-path parity with ``bayes_kit/ess.py``."""
from .diagnostics import ess, ess_imse, ess_ipse  # noqa: F401
