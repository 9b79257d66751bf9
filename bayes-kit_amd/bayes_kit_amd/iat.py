"""Module-path parity with ``bayes_kit/iat.py``."""
from .diagnostics import iat, iat_imse, iat_ipse  # noqa: F401
