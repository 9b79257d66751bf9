"""Convergence diagnostics on the GPU: drop-ins for ``bayes_kit/rhat.py``, ``ess.py``,
``iat.py`` and ``autocorr.py``, plus the streaming / multi-GPU forms the many-chain engine
uses.

Inputs.  Every function accepts what the reference accepts (a 1-D chain, or a sequence of
1-D chains for the R-hat family) and, additionally, a 2-D float64 device tensor ``[N, C]``
(draw-major: N draws of C chains, one chain per GPU lane).  For a 2-D input the per-chain
functions (ess, iat, autocorr) return one value (or column) per chain.

Across GPUs.  Chains are sharded; the only cross-rank traffic is the per-dimension partial
sums of R-hat (a few doubles per dimension).  They are exchanged with ONE small
``all_gather`` per pass (RCCL when the tensors live on the GPU, gloo in the CPU tests) and
combined in rank order, so the result does not depend on the reduction order.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import dist as _bkdist


def _ops(ops):
    return ops if ops is not None else _lib.default_ops()


# ---------------------------------------------------------------------------------------------
# cross-rank combine
# ---------------------------------------------------------------------------------------------
def _gather_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of `t` over the ranks of `group`, accumulated in rank order (deterministic).
    No-op without an initialised process group."""
    from .dist import gather_sum

    return gather_sum(t, group)


def rhat_from_moments(mean_dc, m2_dc, n: int, ops=None, group=None) -> np.ndarray:
    """Per-dimension R-hat (bayes_kit/rhat.py:163-171) from per-chain Welford moments
    ``mean, M2`` ([D, C_local] each, n draws per chain), over ALL ranks' chains.

    Two passes, like ``np.var(means, ddof=1)``: first sum(mean) and sum(var) -> grand mean;
    then sum((mean - grand mean)^2).  Each pass is one kernel + one tiny all_gather.
    """
    ops = _ops(ops)
    D, C = mean_dc.shape
    dev = mean_dc.device
    if n < 2:
        raise ValueError("rhat requires len(chain) >= 2 for every chain in chains")
    part = torch.zeros(3 * D + 1, dtype=torch.float64, device=dev)
    ops.rhat_partials(mean_dc, m2_dc, n, None, part)
    part[3 * D] = float(C)
    tot = _gather_sum(part, group)
    M = float(tot[3 * D].item())
    if M < 2:
        raise ValueError(f"rhat requires len(chains) >= 2, but len(chains) = {int(M)}")
    centre = (tot[0:D] / M).contiguous()
    mean_var = tot[D:2 * D] / M
    part2 = torch.zeros(3 * D + 1, dtype=torch.float64, device=dev)
    ops.rhat_partials(mean_dc, m2_dc, n, centre, part2)
    tot2 = _gather_sum(part2, group)
    var_means = tot2[2 * D:3 * D] / (M - 1.0)
    nf = float(n)
    return torch.sqrt((nf - 1.0) / nf + var_means / mean_var).cpu().numpy()


def _as_dc(theta, D: int, C: int, layout=None):
    """A sampler's draw as a logical [D, C] tensor.  ``sample()`` returns the (C, D) transpose view of the
    engine's [D, C] buffer.  For C != D the shape tells the two apart; for a square draw the strides do
    (the (C, D) view of a chain-contiguous buffer has stride(0) == 1, the buffer itself stride(1) == 1),
    and ``layout`` ("dc" or "cd") overrides both."""
    if theta.dim() != 2:
        raise ValueError(f"expected a 2-D draw, got shape {tuple(theta.shape)}")
    if layout not in (None, "dc", "cd"):
        raise ValueError("layout must be 'dc' ([D, C]) or 'cd' ((C, D))")
    shape = tuple(theta.shape)
    if layout is None:
        if C != D:
            if shape not in ((D, C), (C, D)):
                raise ValueError(f"draw of shape {shape} is neither [{D}, {C}] nor ({C}, {D})")
            layout = "dc" if shape == (D, C) else "cd"
        elif shape != (D, C):
            raise ValueError(f"draw of shape {shape}, expected ({C}, {D})")
        elif theta.stride(1) == 1 and theta.stride(0) != 1:
            layout = "dc"
        elif theta.stride(0) == 1 and theta.stride(1) != 1:
            layout = "cd"
        elif D == 1:
            layout = "dc"
        else:
            raise ValueError(f"a square {shape} draw with strides {tuple(theta.stride())}: pass layout='cd' for sample()'s "
                             "(C, D) or layout='dc' for a [D, C] buffer")
    want = (D, C) if layout == "dc" else (C, D)
    if shape != want:
        raise ValueError(f"draw of shape {shape} with layout={layout!r}, expected {want}")
    return theta if layout == "dc" else theta.t()


class RunningMoments:
    """Streaming per-chain mean / M2 of every dimension (Welford), fed one draw at a time."""

    def __init__(self, D: int, C: int, ops=None):
        self._ops = _ops(ops)
        dev = self._ops.device
        self.mean = torch.zeros((D, C), dtype=torch.float64, device=dev)
        self.m2 = torch.zeros((D, C), dtype=torch.float64, device=dev)
        self.n = 0

    def update(self, theta, layout=None) -> None:
        """theta: the sampler's draw, either the engine's [D, C] buffer or the (C, D) view
        returned by ``sample()`` (told apart by shape, for C == D by strides: see _as_dc)."""
        t = _as_dc(theta, self.mean.shape[0], self.mean.shape[1], layout)
        if (t.shape[1] > 1 and t.stride(1) != 1) or t.dtype != torch.float64:
            t = t.to(torch.float64).contiguous()  # (chains must be contiguous; theta's row pitch may differ from the moments')
        self.n += 1
        self._ops.welford_update(self.mean, self.m2, t, self.n)

    def _update_dev(self, theta_dc, n_dev, n_offset) -> None:
        """update() as a part of a sampler's draw (DrGhmcDiag.attach): the update count is read on the device,
        n = n_dev[0] - n_offset; the caller keeps self.n in step."""
        self._ops.welford_update_dev(self.mean, self.m2, theta_dc, n_dev, n_offset)

    def _update_job(self, n_offset):
        """_update_dev(...) as the Welford part of a job the NEXT draw's generator launch carries along
        (DrGhmcDiag.advance(n); ops.diag_job)."""
        return (self.mean, self.m2, n_offset)

    def rhat(self, group=None) -> np.ndarray:
        return rhat_from_moments(self.mean, self.m2, self.n, self._ops, group)

    # checkpoint / resume: the Welford state is (n, mean, M2); restored, the stream of
    # updates continues bit for bit
    def state_dict(self):
        return {"n": self.n, "mean": self.mean.detach().cpu().clone(), "m2": self.m2.detach().cpu().clone()}

    def load_state_dict(self, sd):
        if tuple(sd["mean"].shape) != tuple(self.mean.shape):
            raise ValueError(f"checkpoint holds moments of shape {tuple(sd['mean'].shape)}, expected {tuple(self.mean.shape)}")
        self.mean.copy_(sd["mean"].to(self.mean.device))
        self.m2.copy_(sd["m2"].to(self.m2.device))
        self.n = int(sd["n"])


class DrawRecorder:
    """Draw storage for post-processing: the full series of a few tracked
    coordinates (and the returned log density) as ``[K, N, C]`` -- draw-major, chain-contiguous,
    the layout ``ess`` / ``rhat`` consume directly -- filled one draw at a time on the device.
    Storing every coordinate of every draw is rarely wanted at 65,536 chains x 1024 dims
    (512 MiB per draw); ``RunningMoments`` covers all coordinates in O(1) memory."""

    def __init__(self, dims: Sequence[int], capacity: int, chains: int, with_logp: bool = True, ops=None):
        self._ops = _ops(ops)
        self.dims = [int(d) for d in dims]
        self.with_logp = bool(with_logp)
        K = len(self.dims) + (1 if with_logp else 0)
        self.series = torch.empty((K, int(capacity), int(chains)), dtype=torch.float64, device=self._ops.device)
        self._dims_dev = torch.tensor(self.dims, dtype=torch.int32, device=self._ops.device)
        self.n = 0

    def _check_dims(self, D: int) -> None:
        """Tracked coordinates against the draw's D: negative indices count from the end (as ``theta[:, d]`` did),
        anything outside [-D, D) raises IndexError -- the kernel reads theta[dims[k] * ld + c] unchecked."""
        if getattr(self, "_dims_D", None) == D:
            return
        norm = []
        for d in self.dims:
            if not -D <= d < D:
                raise IndexError(f"tracked coordinate {d} is out of range for draws with {D} dimensions")
            norm.append(d + D if d < 0 else d)
        self._dims_dev = torch.tensor(norm, dtype=torch.int32, device=self._ops.device)
        self._dims_norm, self._dims_D = norm, D

    def record(self, theta, logp=None, layout=None) -> None:
        """theta: the (C, D) draw returned by ``sample()`` or the [D, C] buffer behind it (told apart by shape, a
        square draw by its strides or by ``layout`` = "cd" / "dc": diagnostics._as_dc); logp: its (C,) log density.
        One launch (bk_record_series) for all tracked series."""
        if self.n >= self.series.shape[1]:
            raise IndexError("DrawRecorder is full")
        C = self.series.shape[2]
        if theta.dim() != 2:
            raise ValueError(f"expected a 2-D draw, got shape {tuple(theta.shape)}")
        if layout == "dc" or (layout is None and theta.shape[1] == C and theta.shape[0] != C):
            D = theta.shape[0]
        else:
            D = theta.shape[1]
        t = _as_dc(theta, D, C, layout)
        self._check_dims(D)
        lp = None
        if self.with_logp:
            lp = logp if (logp.dtype == torch.float64 and logp.is_contiguous()) else logp.to(torch.float64).contiguous()
        if (C == 1 or t.stride(1) == 1) and t.dtype == torch.float64:
            self._ops.record_series(t, self._dims_dev, lp, self.series, self.n)
        else:  # a draw in some other layout / dtype: one strided copy per series
            for k, d in enumerate(self._dims_norm):
                self.series[k, self.n].copy_(t[d])
            if self.with_logp:
                self.series[-1, self.n].copy_(logp)
        self.n += 1

    def _record_dev(self, theta_dc, logp, row_dev, row_offset) -> None:
        """record() as a part of a sampler's draw (DrGhmcDiag.attach): row = row_dev[0] - row_offset, read on the
        device; the caller keeps self.n in step."""
        self._check_dims(theta_dc.shape[0])
        self._ops.record_series_dev(theta_dc, self._dims_dev, logp if self.with_logp else None, self.series, row_dev,
                                    row_offset)

    def _record_job(self, D, logp, row_offset):
        """_record_dev(...) as the tracked-series part of a job the NEXT draw's generator launch carries along."""
        self._check_dims(D)
        return (self._dims_dev, logp if self.with_logp else None, self.series, row_offset)

    def names(self):
        return [f"theta[{d}]" for d in self.dims] + (["logp"] if self.with_logp else [])

    def view(self, k: int) -> torch.Tensor:
        """[n, C] series of tracked quantity k."""
        return self.series[k, : self.n]

    def ess(self) -> torch.Tensor:
        """[K, C] effective sample sizes (ess.py:52-69) of every tracked series of every chain."""
        return torch.stack([ess(self.view(k), ops=self._ops) for k in range(self.series.shape[0])])

    def rhat(self, group=None) -> np.ndarray:
        """R-hat (rhat.py:111-171) of every tracked quantity over all chains (and ranks)."""
        return np.array([rhat(self.view(k), ops=self._ops, group=group) for k in range(self.series.shape[0])])

    def state_dict(self):
        return {"series": self.series[:, : self.n].cpu().clone(), "dims": self.dims, "with_logp": self.with_logp,
                "n": self.n}

    def load_state_dict(self, sd):
        if list(sd["dims"]) != self.dims or bool(sd["with_logp"]) != self.with_logp:
            raise ValueError(f"checkpoint tracks dims {sd['dims']} (logp: {sd['with_logp']}), this recorder {self.dims} "
                             f"(logp: {self.with_logp})")
        ser = sd["series"]
        if ser.shape[0] != self.series.shape[0] or ser.shape[2] != self.series.shape[2] or ser.shape[1] > self.series.shape[1]:
            raise ValueError(f"checkpoint series {tuple(ser.shape)} does not fit the recorder {tuple(self.series.shape)}")
        self.n = int(ser.shape[1])
        self.series[:, : self.n].copy_(ser.to(self.series.device))


class DrawStore:
    """Chunked draw storage: every coordinate of every draw of this rank's chains, as
    fixed-size chunks ``[draws, D, C]`` -- draw-major, chain-contiguous: the series of coordinate d
    of a chunk is the strided view ``chunk[:, d, :]`` (``[n, C]``, chains contiguous), which ``ess`` /
    ``rhat`` / ``autocorr`` consume as it is, no reshaping.  Draws are staged in a device buffer and
    written one ``chunk_#####.npy`` file per full chunk (plus ``meta.json``); one directory per rank.

        store = DrawStore.create(path, D, C, chunk=64)        # writer
        store.append(theta)  ...  store.close()
        store = DrawStore.open(path)                           # reader
        x = store.series(d)        # [N, C] device tensor      r = rhat(x); e = ess(x)

    A store can be re-opened for appending (``DrawStore.create(..., resume=True)``): with the
    sampler's ``state_dict`` this makes a run restartable at any chunk boundary or in between (the
    partial chunk is kept in ``state_dict()`` of the store).
    """

    def __init__(self, path, D, C, chunk, ops, chunks):
        import os

        self.path, self.D, self.C, self.chunk = str(path), int(D), int(C), int(chunk)
        self._ops = _ops(ops)
        self._os = os
        self._chunks = list(chunks)  # draws per completed chunk file
        self._buf_t = None           # [chunk, D, C] device staging buffer: allocated by the first append()
        self._fill = 0

    @property
    def _buf(self):
        # a reader (DrawStore.open) never appends and never pays for the staging buffer: at config-3
        # size and chunk = 64 it would be 34 GB of HBM
        if self._buf_t is None:
            self._buf_t = torch.empty((self.chunk, self.D, self.C), dtype=torch.float64, device=self._ops.device)
        return self._buf_t

    # -- writer -----------------------------------------------------------------------------------
    @classmethod
    def create(cls, path, D, C, chunk=64, ops=None, resume=False):
        import json
        import os

        os.makedirs(path, exist_ok=True)
        meta = os.path.join(path, "meta.json")
        chunks = []
        if os.path.exists(meta):
            if not resume:
                raise FileExistsError(f"{path} already holds a draw store (pass resume=True to append)")
            with open(meta) as f:
                m = json.load(f)
            if (m["D"], m["C"], m["chunk"]) != (int(D), int(C), int(chunk)):
                raise ValueError(f"{path} holds a store of D={m['D']}, C={m['C']}, chunk={m['chunk']}")
            chunks = m["chunks"]
        st = cls(path, D, C, chunk, ops, chunks)
        st._write_meta()
        return st

    def _write_meta(self):
        import json

        with open(self._os.path.join(self.path, "meta.json"), "w") as f:
            json.dump({"D": self.D, "C": self.C, "chunk": self.chunk, "chunks": self._chunks, "dtype": "float64",
                       "layout": "[draw, D, C], C contiguous"}, f)

    def append(self, theta, layout=None) -> None:
        """theta: the sampler's draw -- the (C, D) view returned by ``sample()`` or a [D, C] buffer (told
        apart by shape; for C == D by strides, or say ``layout="cd"`` / ``"dc"``)."""
        t = _as_dc(theta, self.D, self.C, layout)
        self._buf[self._fill].copy_(t)
        self._fill += 1
        if self._fill == self.chunk:
            self._flush()

    def _flush(self):
        if self._fill == 0:
            return
        arr = self._buf[: self._fill].cpu().numpy()
        np.save(self._os.path.join(self.path, "chunk_%05d.npy" % len(self._chunks)), arr)
        self._chunks.append(int(self._fill))
        self._fill = 0
        self._write_meta()

    def close(self):
        self._flush()

    def state_dict(self):
        """The not yet written part of the current chunk (what a checkpoint must carry besides the files)."""
        part = self._buf[: self._fill].cpu().clone() if self._fill else torch.empty((0, self.D, self.C), dtype=torch.float64)
        return {"chunks": list(self._chunks), "partial": part}

    def load_state_dict(self, sd, truncate=False):
        """Rewind the store to a checkpoint.  Chunk files written AFTER the checkpoint belong to draws the
        resumed run will produce again: with ``truncate=True`` they are deleted, otherwise they are kept on
        disk under ``superseded_<k>_chunk_#####.npy`` (never read again by this store) -- nothing is removed
        unless asked for."""
        if len(sd["chunks"]) > len(self._chunks) or sd["chunks"] != self._chunks[: len(sd["chunks"])]:
            raise ValueError("the checkpoint refers to chunk files this store does not hold")
        for k in range(len(sd["chunks"]), len(self._chunks)):
            f = self._os.path.join(self.path, "chunk_%05d.npy" % k)
            if truncate:
                self._os.remove(f)
            else:
                gen = 0
                while self._os.path.exists(self._os.path.join(self.path, "superseded_%d_chunk_%05d.npy" % (gen, k))):
                    gen += 1
                self._os.replace(f, self._os.path.join(self.path, "superseded_%d_chunk_%05d.npy" % (gen, k)))
        self._chunks = list(sd["chunks"])
        part = sd["partial"]
        self._fill = int(part.shape[0])
        if self._fill:
            self._buf[: self._fill].copy_(part.to(self._ops.device))
        self._write_meta()

    # -- reader -----------------------------------------------------------------------------------
    @classmethod
    def open(cls, path, ops=None):
        import json
        import os

        with open(os.path.join(path, "meta.json")) as f:
            m = json.load(f)
        return cls(path, m["D"], m["C"], m["chunk"], ops, m["chunks"])

    @property
    def draws(self) -> int:
        return sum(self._chunks) + self._fill

    def chunk_array(self, k) -> np.ndarray:
        """Chunk k as a memory-mapped [n, D, C] array."""
        return np.load(self._os.path.join(self.path, "chunk_%05d.npy" % k), mmap_mode="r")

    def series(self, d: int) -> torch.Tensor:
        """All draws of coordinate d: [N, C] device tensor (each chunk contributes its [n, C] slice; only
        that slice is read from disk)."""
        N = self.draws
        out = torch.empty((N, self.C), dtype=torch.float64, device=self._ops.device)
        pos = 0
        for k, n in enumerate(self._chunks):
            out[pos:pos + n].copy_(torch.from_numpy(np.ascontiguousarray(self.chunk_array(k)[:, d, :])))
            pos += n
        if self._fill:
            out[pos:pos + self._fill].copy_(self._buf[: self._fill, d, :])
        return out

    def rhat(self, dims=None, group=None) -> np.ndarray:
        dims = range(self.D) if dims is None else dims
        return np.array([rhat(self.series(d), ops=self._ops, group=group) for d in dims])

    def ess(self, dims=None) -> torch.Tensor:
        dims = range(self.D) if dims is None else dims
        return torch.stack([ess(self.series(d), ops=self._ops) for d in dims])


# ---------------------------------------------------------------------------------------------
# helpers for the reference-style inputs
# ---------------------------------------------------------------------------------------------
def _is_matrix(x) -> bool:
    return isinstance(x, torch.Tensor) and x.dim() == 2


def _pack_chains(chains: Sequence, ops):
    """Sequence of (possibly ragged) 1-D chains -> ([Nmax, M] device tensor, lengths)."""
    arrs = [np.asarray(c, dtype=np.float64).reshape(-1) for c in chains]
    lens = np.array([a.shape[0] for a in arrs], dtype=np.int32)
    nmax = int(lens.max()) if len(arrs) else 0
    host = np.zeros((max(nmax, 1), len(arrs)), dtype=np.float64)
    for j, a in enumerate(arrs):
        host[: a.shape[0], j] = a
    return torch.from_numpy(host).to(ops.device), lens


def _rhat_of_columns(x, lens, ops, group=None):
    """R-hat of the chains stored as columns of x (lens None = all rows)."""
    N, M = x.shape
    dev = x.device
    mean = torch.empty((1, M), dtype=torch.float64, device=dev)
    var = torch.empty((1, M), dtype=torch.float64, device=dev)
    lt = None if lens is None else torch.from_numpy(np.asarray(lens, dtype=np.int32)).to(dev)
    ops.chain_mean_var(x, lt, mean[0], var[0])
    # reuse the per-dimension partial-sum kernel with D = 1 (n = 2 makes M2/(n-1) = var)
    part = torch.zeros(4, dtype=torch.float64, device=dev)
    ops.rhat_partials(mean, var, 2, None, part)
    part[3] = float(M)
    len_sum = torch.tensor([float(N * M) if lens is None else float(np.sum(lens))], dtype=torch.float64, device=dev)
    tot = _gather_sum(torch.cat([part, len_sum]), group)
    Mt = float(tot[3].item())
    if Mt < 2:  # rhat.py:159-160, over all ranks' chains
        raise ValueError(f"rhat requires len(chains) >= 2, but len(chains) = {int(Mt)}")
    centre = (tot[0:1] / Mt).contiguous()
    part2 = torch.zeros(4, dtype=torch.float64, device=dev)
    ops.rhat_partials(mean, var, 2, centre, part2)
    tot2 = _gather_sum(part2, group)
    nbar = float(tot[4].item()) / Mt
    var_means = float(tot2[2].item()) / (Mt - 1.0)
    mean_vars = float(tot[1].item()) / Mt
    return np.float64(np.sqrt((nbar - 1.0) / nbar + var_means / mean_vars))


# ---------------------------------------------------------------------------------------------
# rhat.py
# ---------------------------------------------------------------------------------------------
def rhat(chains, *, ops=None, group=None):
    """Potential scale reduction factor, bayes_kit/rhat.py:111-171.

    ``chains``: a sequence of 1-D chains (ragged allowed) or a device tensor [N, C].  With a
    process group, ``chains`` is this rank's shard and the statistic is over all ranks.
    """
    ops = _ops(ops)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if _is_matrix(chains):
        N, M = chains.shape
        if world == 1 and M < 2:
            raise ValueError(f"rhat requires len(chains) >= 2, but len(chains) = {M}")
        if N < 2:
            raise ValueError("rhat requires len(chain) >= 2 for every chain in chains")
        x = chains if chains.stride(1) == 1 else chains.contiguous()
        return _rhat_of_columns(x, None, ops, group)
    if world == 1 and len(chains) < 2:
        raise ValueError(f"rhat requires len(chains) >= 2, but {len(chains) = }")
    if not all(len(chain) >= 2 for chain in chains):
        raise ValueError("rhat requires len(chain) >= 2 for every chain in chains")
    x, lens = _pack_chains(chains, ops)
    return _rhat_of_columns(x, lens, ops, group)


def split_chains(chains):
    """bayes_kit/rhat.py:9-24 (host-side index arithmetic only: no numerics involved)."""
    return [arr for chain in chains for arr in np.array_split(chain, 2)]


def split_rhat(chains, *, ops=None, group=None):
    """bayes_kit/rhat.py:174-202: R-hat of the chains split in half (first half one longer
    for odd lengths)."""
    ops = _ops(ops)
    if _is_matrix(chains):
        N, M = chains.shape
        x = chains if chains.stride(1) == 1 else chains.contiguous()
        h = (N + 1) // 2
        if N - h < 2:
            raise ValueError("rhat requires len(chain) >= 2 for every chain in chains")
        halves = torch.zeros((h, 2 * M), dtype=torch.float64, device=x.device)
        halves[:, :M] = x[:h]
        halves[: N - h, M:] = x[h:]
        lens = np.concatenate([np.full(M, h), np.full(M, N - h)]).astype(np.int32)
        return _rhat_of_columns(halves, lens, ops, group)
    return rhat(split_chains(chains), ops=ops, group=group)


def _canonical_keys(flat: torch.Tensor) -> torch.Tensor:
    """Sort keys whose RADIX order (the order of the raw bit patterns, which is what bk_sort_by_key and the
    splitter search see) is numpy's comparison order (rhat.py:51-52): -0.0 becomes +0.0, so that the two
    zeros tie and rank in pooled order, and every NaN becomes the positive quiet NaN, which sorts last
    as in ``np.argsort`` (a sign-bit NaN would sort first)."""
    flat = flat + 0.0  # (-0.0) + (+0.0) = +0.0 in round-to-nearest; every other value unchanged
    return torch.where(torch.isnan(flat), torch.full_like(flat, float("nan")), flat)


def _ranks_pooled(flat: torch.Tensor, ops) -> torch.Tensor:
    """Ascending 1-based ranks of the pooled draws (rhat.py:51-52: argsort().argsort() + 1).
    Ties get distinct consecutive ranks; numpy leaves their order to its sort, here it is the
    order of appearance (stable sort).  bk_sort_by_key + bk_scatter_ranks."""
    n = flat.numel()
    idx = torch.arange(n, dtype=torch.int64, device=flat.device)
    _, payload = ops.sort_by_key(_canonical_keys(flat).contiguous(), idx)
    ranks = torch.empty_like(flat)
    ops.scatter_ranks(payload, 0.0, ranks)
    return ranks


SAMPLES_PER_RANK = 64  # regular samples each rank contributes per destination bucket


def _ranks_pooled_across_ranks(x: torch.Tensor, ops, group=None) -> torch.Tensor:
    """Global pooled ranks of this rank's [N, C_local] shard WITHOUT replicating the draws: a sample
    sort.  Pooled order = chain-major over all ranks' chains in rank order (what rank_chains sees
    when handed every chain); equal values rank in that order.

      1. local stable sort of (value, global flat index)                       bk_sort_by_key
      2. 64*world regular samples per rank -> all_gather -> world-1 splitters
      3. bucket boundaries by binary search (bk_count_below); bucket k belongs to rank k
      4. all_to_all of the buckets (values + indices): S_local*16 bytes sent per rank, over RCCL/xGMI,
         instead of S_total*8 bytes RECEIVED by every rank
      5. owner: stable sort of what arrived (sorted runs in source-rank order: ties stay in global
         order); rank of position j = (sizes of the lower buckets) + j + 1      bk_scatter_ranks
      6. ranks travel back to the elements' home ranks (all_to_all)
    Returns the ranks as doubles, shaped like x."""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    N, C = x.shape
    dev = x.device
    flat = _canonical_keys(x.t().contiguous().reshape(-1))
    S_local = flat.numel()
    sizes = _all_gather_counts(S_local, dev, group)
    ends = torch.cumsum(torch.tensor(sizes, dtype=torch.int64, device=dev), 0)
    first = int(ends[me].item()) - S_local
    gidx = torch.arange(S_local, dtype=torch.int64, device=dev) + first
    keys, pay = ops.sort_by_key(flat, gidx)
    # 2. splitters
    s = SAMPLES_PER_RANK * world
    samp = torch.full((s,), float("inf"), dtype=torch.float64, device=dev)
    if S_local > 0:
        k = min(s, S_local)
        pos = ((torch.arange(k, dtype=torch.float64, device=dev) + 0.5) * (S_local / k)).to(torch.int64).clamp_(max=S_local - 1)
        samp[:k] = keys[pos]
    parts = _bkdist.all_gather(samp, group)
    cat = torch.cat(parts)                                  # world*s values: tiny
    allsamp, _ = ops.sort_by_key(cat, torch.arange(cat.numel(), dtype=torch.int64, device=dev))
    nreal = int(torch.isfinite(allsamp).sum().item())
    cut_pos = [(nreal * (r + 1)) // world for r in range(world - 1)]
    splitters = allsamp[torch.tensor(cut_pos, dtype=torch.int64, device=dev).clamp_(max=max(nreal - 1, 0))]
    # 3. buckets (values equal to a splitter go to the bucket above it, on every rank alike)
    cuts = ops.count_below(keys, splitters.contiguous()) if world > 1 else torch.empty(0, dtype=torch.int64, device=dev)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cuts, torch.tensor([S_local], dtype=torch.int64, device=dev)])
    bounds = torch.cummax(bounds, 0).values                # (monotone even if splitters repeat)
    send_counts = (bounds[1:] - bounds[:-1])
    recv_counts = torch.empty_like(send_counts)
    _bkdist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    # 4. the buckets travel
    R = sum(rc)
    rk = torch.empty(R, dtype=torch.float64, device=dev)
    rp = torch.empty(R, dtype=torch.int64, device=dev)
    _bkdist.all_to_all_single(rk, keys, rc, sc, group=group)
    _bkdist.all_to_all_single(rp, pay, rc, sc, group=group)
    # 5. ranks inside my bucket
    _, ps = ops.sort_by_key(rk, rp)
    bsz = _all_gather_counts(R, dev, group)
    my_base = float(sum(bsz[:me]))
    # 6. back home: group the sorted positions by the rank that owns the element
    owner = torch.searchsorted(ends, ps, right=True)
    _, order = ops.sort_by_key(owner.to(torch.float64), torch.arange(R, dtype=torch.int64, device=dev))
    back_idx = ps[order].contiguous()
    back_rank = (my_base + (order + 1).to(torch.float64)).contiguous()
    hi = torch.empty(S_local, dtype=torch.int64, device=dev)
    hr = torch.empty(S_local, dtype=torch.float64, device=dev)
    _bkdist.all_to_all_single(hi, back_idx, sc, rc, group=group)
    _bkdist.all_to_all_single(hr, back_rank, sc, rc, group=group)
    ranks = torch.empty(S_local, dtype=torch.float64, device=dev)
    ranks[hi - first] = hr
    return ranks.reshape(C, N).t()


def _pooled(chains, ops):
    """-> (flat chain-major device vector, list of lengths)."""
    if _is_matrix(chains):
        N, M = chains.shape
        return chains.t().contiguous().reshape(-1), [N] * M
    arrs = [np.asarray(c, dtype=np.float64).reshape(-1) for c in chains]
    flat = np.concatenate(arrs) if arrs else np.zeros(0)
    return torch.from_numpy(flat).to(ops.device), [a.shape[0] for a in arrs]


def rank_chains(chains, *, ops=None):
    """bayes_kit/rhat.py:27-59: chains with every value replaced by its pooled rank."""
    if len(chains) == 0:
        return chains
    ops = _ops(ops)
    flat, lens = _pooled(chains, ops)
    ranks = _ranks_pooled(flat, ops)
    if _is_matrix(chains):
        return ranks.reshape(len(lens), lens[0]).t()
    out, pos = [], 0
    r = ranks.cpu().numpy()
    for n in lens:
        out.append(r[pos:pos + n])
        pos += n
    return out


def rank_normalize_chains(chains, *, ops=None):
    """bayes_kit/rhat.py:62-108: Phi^-1((rank - 0.325) / (S - 0.25)) (the offset the reference
    actually uses, not the 3/8 of its docstring)."""
    ops = _ops(ops)
    flat, lens = _pooled(chains, ops)
    S = flat.numel()
    z = torch.empty_like(flat)
    ops.rank_normalize(_ranks_pooled(flat, ops), float(S), z)
    if _is_matrix(chains):
        return z.reshape(len(lens), lens[0]).t().contiguous()
    out, pos = [], 0
    zz = z.cpu().numpy()
    for n in lens:
        out.append(zz[pos:pos + n])
        pos += n
    return out


def _all_gather_counts(n: int, device, group=None):
    """Every rank's value of the integer n, in rank order (dist.shard hands the remainder of an
    uneven split to the first ranks, so shard widths may differ by one)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
    return [int(p.item()) for p in _bkdist.all_gather(mine, group)]


def _all_gather_columns(x: torch.Tensor, group=None):
    """[N, C_local] on every rank -> ([N, C_total] in rank order, first column of this rank).
    C_local may differ between ranks: shards are padded to the widest for the collective and
    the padding is dropped afterwards."""
    world = dist.get_world_size(group)
    counts = _all_gather_counts(x.shape[1], x.device, group)
    cmax = max(counts)
    send = x.contiguous()
    if x.shape[1] != cmax:
        send = torch.zeros((x.shape[0], cmax), dtype=x.dtype, device=x.device)
        send[:, : x.shape[1]] = x
    parts = _bkdist.all_gather(send, group)
    full = torch.cat([p[:, :c] for p, c in zip(parts, counts)], dim=1)
    return full, sum(counts[: dist.get_rank(group)])


def rank_normalized_rhat(chains, *, ops=None, group=None):
    """bayes_kit/rhat.py:205-236: split R-hat of the rank-normalised chains.

    Across ranks (an [N, C_local] shard per rank) the ranks are global: they come from a sample
    sort over the process group (_ranks_pooled_across_ranks: each rank sends and receives its own
    share of the draws once, nothing is replicated); every rank then turns its own chains' ranks
    into normal scores and joins the usual cross-rank split R-hat."""
    multi = _bkdist.collectives_active(group)
    if not multi:
        return split_rhat(rank_normalize_chains(chains, ops=ops), ops=ops, group=group)  # (a one-rank group stays local)
    if not _is_matrix(chains):
        raise ValueError("across ranks rank_normalized_rhat takes this rank's [N, C_local] device tensor")
    ops = _ops(ops)
    ranks = _ranks_pooled_across_ranks(chains, ops, group).contiguous()
    S = float(sum(_all_gather_counts(chains.numel(), chains.device, group)))
    z = torch.empty_like(ranks)
    ops.rank_normalize(ranks, S, z)
    return split_rhat(z, ops=ops, group=group)


# ---------------------------------------------------------------------------------------------
# autocorr.py / iat.py / ess.py
# ---------------------------------------------------------------------------------------------
def _series(chain, ops):
    """-> ([N, C] device tensor, was_1d)."""
    if _is_matrix(chain):
        return (chain if chain.stride(1) == 1 else chain.contiguous()), False
    a = np.asarray(chain, dtype=np.float64).reshape(-1)
    return torch.from_numpy(a.copy()).to(ops.device).reshape(-1, 1), True


def _len(chain) -> int:
    return chain.shape[0] if _is_matrix(chain) else len(chain)


# Chains up to FFT_MIN_DRAWS draws: series staged in LDS, direct lag sums (N steps per 64 lags, only the lags the
# Geyer truncation needs).  Longer chains: the reference's own FFT formulation (autocorr.py:26-32), O(N log N)
# per chain, by the library's Stockham passes over the [N, C] layout (bk_autocorr_fft: one lane per pair of
# chains, two real series per complex transform), followed by the scan kernel (bk_iat_from_acor) -- the direct
# sums are O(N^2 / 64) for ALL lags.
FFT_MIN_DRAWS = 16384
# autocorr() wants ALL lags, which the direct sums pay O(N^2 / 64) for: the FFT is ahead from 256 draws on (4,096 chains:
# 0.10 vs 0.13 ms at 256 draws, 0.40 vs 0.66 at 2,048, 0.55 vs 3.2 at 8,192; tools/attic/autocorr_crossover.py)
AUTOCORR_FFT_MIN_DRAWS = 256
_FFT_SCRATCH_BYTES = 2 << 30  # chains per FFT batch are chosen to keep the two complex scratch arrays below this


def _autocorr_fft(x: torch.Tensor, ops) -> torch.Tensor:
    """autocorr.py:23-33 for every column of x [N, C]."""
    N, C = x.shape
    size = 1 << int(np.ceil(np.log2(2 * N - 1)))
    block = max(128, (_FFT_SCRATCH_BYTES // (size * 16)) // 128 * 128)  # (2 buffers x size x block/2 x 16 B)
    need = getattr(ops, "autocorr_fft_work_bytes", None)
    while need is not None and block > 128 and need(N, min(block, C)) > 2 * _FFT_SCRATCH_BYTES:
        block = max(128, block // 2 // 128 * 128)  # (the plan pads its rows: size the batch by what it will really take)
    out = torch.empty_like(x)
    for c0 in range(0, C, block):
        ops.autocorr_fft(x[:, c0:c0 + block], out[:, c0:c0 + block])
    return out


def autocorr(chain, *, ops=None):
    """Autocorrelation at all lags, bayes_kit/autocorr.py:6-33."""
    if _len(chain) < 2:
        raise ValueError(f"autocorr requires len(chain) >= 2, but {len(chain)=}")
    ops = _ops(ops)
    x, one = _series(chain, ops)
    if x.shape[0] >= AUTOCORR_FFT_MIN_DRAWS:
        out = _autocorr_fft(x, ops)
    else:
        out = torch.empty_like(x)
        ops.autocorr(x, out)
    return out[:, 0].cpu().numpy() if one else out


def _iat_ess(chain, estimator, want, ops):
    ops = _ops(ops)
    x, one = _series(chain, ops)
    N, C = x.shape
    ess_out = torch.empty(C, dtype=torch.float64, device=x.device)
    iat_out = torch.empty(C, dtype=torch.float64, device=x.device)
    if N >= FFT_MIN_DRAWS:
        ops.iat_from_acor(_autocorr_fft(x, ops), estimator, ess_out, iat_out)
    else:
        ops.ess(x, estimator, ess_out, iat_out)
    r = ess_out if want == "ess" else iat_out
    return np.float64(r[0].item()) if one else r


def iat_ipse(chain, *, ops=None):
    """bayes_kit/iat.py:46-92."""
    if _len(chain) < 4:
        raise ValueError(f"ess requires len(chains) >= 4, but {len(chain)=}")
    return _iat_ess(chain, 1, "iat", ops)


def iat_imse(chain, *, ops=None):
    """bayes_kit/iat.py:95-135."""
    if _len(chain) < 4:
        raise ValueError(f"iat requires len(chains) >=4, but {len(chain) = }")
    return _iat_ess(chain, 0, "iat", ops)


def iat(chain, *, ops=None):
    """bayes_kit/iat.py:138-156."""
    return iat_imse(chain, ops=ops)


def ess_ipse(chain, *, ops=None):
    """bayes_kit/ess.py:5-21."""
    if _len(chain) < 4:
        raise ValueError(f"ess_ipse(chain) requires len(chain) >= 4, but {len(chain)=}")
    return _iat_ess(chain, 1, "ess", ops)


def ess_imse(chain, *, ops=None):
    """bayes_kit/ess.py:24-49."""
    if _len(chain) < 4:
        raise ValueError(f"ess_imse(chain) requires len(chain) >=4, but {len(chain) = }")
    return _iat_ess(chain, 0, "ess", ops)


def ess(chain, *, ops=None):
    """bayes_kit/ess.py:52-69."""
    if _len(chain) < 4:
        raise ValueError(f"ess(chain) requires len(chain) >=4, but {len(chain) = }")
    return _iat_ess(chain, 0, "ess", ops)
