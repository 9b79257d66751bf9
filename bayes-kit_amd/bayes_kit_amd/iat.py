"""Module-path parity with ``bayes_kit/iat.py``."""
from .diagnostics import iat, iat_imse, iat_ipse  # noqa: F401


def _end_pos_pairs(acor, *, ops=None):
    """bayes_kit/iat.py:7-43: index one past the last even-aligned pair of autocorrelations before
    the first pair with a negative sum.  A 1-D sequence gives an int as in the reference; an
    [N, C] device tensor (one column per chain) gives a (C,) int64 tensor."""
    import numpy as np
    import torch

    from . import _lib

    ops = ops if ops is not None else _lib.default_ops()
    if isinstance(acor, torch.Tensor) and acor.dim() == 2:
        a = acor if acor.stride(1) == 1 or acor.shape[1] == 1 else acor.contiguous()
        out = torch.empty(a.shape[1], dtype=torch.int64, device=ops.device)
        ops.end_pos_pairs(a.to(device=ops.device, dtype=torch.float64), out)
        return out
    a = torch.from_numpy(np.asarray(acor, dtype=np.float64).reshape(-1, 1).copy()).to(ops.device)
    out = torch.empty(1, dtype=torch.int64, device=ops.device)
    ops.end_pos_pairs(a, out)
    return int(out[0].item())
