"""``Stretcher`` placeholder, as in the reference: ``bayes_kit/ensemble.py:9-69`` defines
the class with a docstring only (its whole affine-invariant implementation is commented
out and untested), so there is no behaviour to reproduce."""


class Stretcher:
    """Affine-invariant ensemble sampler (not implemented upstream either)."""
