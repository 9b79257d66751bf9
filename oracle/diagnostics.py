"""NumPy restatement of the reference's convergence diagnostics (TEST INFRASTRUCTURE).

Follows ``bayes_kit/rhat.py``, ``ess.py``, ``iat.py``, ``autocorr.py``; pinned by the
literal known answers in the reference's tests and by golden vectors (see
``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import numpy as np
import scipy.stats


# ---- rhat.py -------------------------------------------------------------------
def split_chains(chains):
    """rhat.py:9-24 -- first half one longer for odd lengths."""
    out = []
    for ch in chains:
        out.extend(np.array_split(ch, 2))
    return out


def rank_chains(chains):
    """rhat.py:27-59 -- ascending ranks from 1 over the pooled draws."""
    if len(chains) == 0:
        return chains
    pooled = np.concatenate(chains)
    ranks = (pooled.argsort().argsort() + 1).astype(np.float64)
    out, pos = [], 0
    for ch in chains:
        out.append(ranks[pos : pos + len(ch)])
        pos += len(ch)
    return out


def rank_normalize_chains(chains):
    """rhat.py:62-108 -- note the 0.325 offset actually used at :106."""
    S = sum(len(ch) for ch in chains)
    return [
        [scipy.stats.norm.ppf((r - 0.325) / (S - 0.25)) for r in ranked]
        for ranked in rank_chains(chains)
    ]


def rhat(chains):
    """rhat.py:111-171."""
    if len(chains) < 2:
        raise ValueError(f"rhat requires len(chains) >= 2, but {len(chains) = }")
    if not all(len(ch) >= 2 for ch in chains):
        raise ValueError("rhat requires len(chain) >= 2 for every chain in chains")
    nbar = np.mean([len(ch) for ch in chains])
    means = [np.mean(ch) for ch in chains]
    variances = [np.var(ch, ddof=1) for ch in chains]
    return np.sqrt((nbar - 1) / nbar + np.var(means, ddof=1) / np.mean(variances))


def split_rhat(chains):
    """rhat.py:174-202."""
    return rhat(split_chains(chains))


def rank_normalized_rhat(chains):
    """rhat.py:205-236."""
    return split_rhat(rank_normalize_chains(chains))


# ---- autocorr.py ---------------------------------------------------------------
def autocorr(chain):
    """autocorr.py:6-33 -- FFT autocorrelation, padded to 2**ceil(log2(2N-1))."""
    if len(chain) < 2:
        raise ValueError(f"autocorr requires len(chain) >= 2, but {len(chain)=}")
    chain = np.asarray(chain)
    size = 2 ** np.ceil(np.log2(2 * len(chain) - 1)).astype("int")
    var = np.var(chain)
    centred = chain - np.mean(chain)
    spec = np.fft.fft(centred, size)
    power = np.abs(spec) ** 2
    N = len(centred)
    return (np.fft.ifft(power).real / var / N)[0:N]


# ---- iat.py --------------------------------------------------------------------
def _end_pos_pairs(acor):
    """iat.py:7-43."""
    N = len(acor)
    n = 0
    while n + 1 < N:
        if acor[n] + acor[n + 1] < 0:
            return n
        n += 2
    return n


def iat_ipse(chain):
    """iat.py:46-92."""
    if len(chain) < 4:
        raise ValueError(f"ess requires len(chains) >= 4, but {len(chain)=}")
    acor = autocorr(chain)
    n = _end_pos_pairs(acor)
    return 2 * acor[0:n].sum() - 1


def iat_imse(chain):
    """iat.py:95-135 -- running minimum of the pair sums."""
    if len(chain) < 4:
        raise ValueError(f"iat requires len(chains) >=4, but {len(chain) = }")
    acor = autocorr(chain)
    n = _end_pos_pairs(acor)
    prev_min = acor[0] + acor[1]
    total = prev_min
    i = 2
    while i + 1 < n:
        prev_min = min(prev_min, acor[i] + acor[i + 1])
        total += prev_min
        i += 2
    return 2 * total - 1


def iat(chain):
    """iat.py:138-156."""
    return iat_imse(chain)


# ---- ess.py --------------------------------------------------------------------
def ess_ipse(chain):
    """ess.py:5-21."""
    if len(chain) < 4:
        raise ValueError(f"ess_ipse(chain) requires len(chain) >= 4, but {len(chain)=}")
    return len(chain) / iat_ipse(chain)


def ess_imse(chain):
    """ess.py:24-49."""
    if len(chain) < 4:
        raise ValueError(f"ess_imse(chain) requires len(chain) >=4, but {len(chain) = }")
    return len(chain) / iat_imse(chain)


def ess(chain):
    """ess.py:52-69."""
    if len(chain) < 4:
        raise ValueError(f"ess(chain) requires len(chain) >=4, but {len(chain) = }")
    return len(chain) / iat(chain)
