"""Shared many-chain machinery: chain-contiguous state buffers, per-chain random streams,
and the bridge between the reference's Model protocol and the device.

State layout: every phase-space array is a float64 ``[D, C]`` tensor with the chain index
contiguous (one chain per GPU lane; a wavefront reads 64 consecutive doubles for each d).
A model sees the ``(C, D)`` transpose view with strides ``(1, ld)``.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

_M64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------
# per-chain random streams
# ---------------------------------------------------------------------------------------
def _bitgen_words(bg) -> tuple:
    """(kind, uint64[RNG_WORDS]) for a numpy BitGenerator (Philox or PCG64)."""
    st = bg.state
    w = np.zeros(_lib.RNG_WORDS, dtype=np.uint64)
    name = st["bit_generator"]
    if name == "Philox":
        w[0:2] = st["state"]["key"]
        w[2:6] = st["state"]["counter"]
        w[6:10] = st["buffer"]
        w[10] = st["buffer_pos"]
        return _lib.RNG_PHILOX, w
    if name == "PCG64":
        if st.get("has_uint32", 0):
            raise ValueError("PCG64 generator holds a cached 32-bit half draw; not supported")
        s, inc = st["state"]["state"], st["state"]["inc"]
        w[0], w[1], w[2], w[3] = s >> 64, s & _M64, inc >> 64, inc & _M64
        return _lib.RNG_PCG64, w
    raise TypeError(f"unsupported bit generator {name}; use Philox or PCG64")


def _as_bitgen(seed):
    if isinstance(seed, np.random.Generator):
        return seed.bit_generator
    if isinstance(seed, np.random.BitGenerator):
        return seed
    return None


def make_streams(seed, C: int, chain_id0: int, reference_int_seed: bool, device):
    """Build the per-chain RNG table.

    Returns ``(kind, state)`` with ``state`` an int64 tensor ``[RNG_WORDS, C]`` holding the
    uint64 words of include/bkhip.h.

    * ``None``  -> Philox with a key from ``os.urandom`` (non-reproducible, like
      ``np.random.default_rng(None)``, bayes_kit/hmc.py:23).
    * ``int``   -> chain c uses ``np.random.Philox(key=[seed, chain_id0 + c])``; with
      ``reference_int_seed`` (single-chain drop-in mode) the stream is instead
      ``np.random.default_rng(seed)`` itself, i.e. PCG64, exactly as the reference.
    * a numpy ``BitGenerator``/``Generator`` (Philox or PCG64), or a sequence of C of them:
      the stream continues from that object's current state (the object itself is not
      advanced afterwards).
    """
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")
        reference_int_seed = False
    if isinstance(seed, (int, np.integer)) and not isinstance(seed, bool):
        if reference_int_seed:
            if C != 1:
                raise ValueError("an int seed selects the reference's PCG64 stream only for one chain")
            kind, w = _bitgen_words(np.random.default_rng(int(seed)).bit_generator)
            words = w[:, None]
        else:
            kind = _lib.RNG_PHILOX
            words = np.zeros((_lib.RNG_WORDS, C), dtype=np.uint64)
            words[0, :] = np.uint64(int(seed) & _M64)
            words[1, :] = (np.arange(C, dtype=np.uint64) + np.uint64(chain_id0 & _M64))
            words[10, :] = 4
    else:
        gens: Sequence
        bg = _as_bitgen(seed)
        if bg is not None:
            gens = [bg]
        else:
            gens = [_as_bitgen(s) for s in seed]
            if any(g is None for g in gens):
                raise TypeError("seed must be None, an int, a numpy (Bit)Generator or a sequence of them")
        if len(gens) != C:
            raise ValueError(f"{len(gens)} bit generators given for {C} chains")
        parts = [_bitgen_words(g) for g in gens]
        kinds = {k for k, _ in parts}
        if len(kinds) != 1:
            raise ValueError("all chains must use the same kind of bit generator")
        kind = kinds.pop()
        words = np.stack([w for _, w in parts], axis=1)
    state = torch.from_numpy(np.ascontiguousarray(words).view(np.int64)).to(device)
    return kind, state


def numpy_generator_from_words(kind: int, w: np.ndarray) -> np.random.Generator:
    """Rebuild a numpy Generator positioned exactly where a device stream is."""
    w = [int(v) for v in w]
    if kind == _lib.RNG_PHILOX:
        bg = np.random.Philox()
        st = bg.state
        st["state"]["key"] = np.array(w[0:2], dtype=np.uint64)
        st["state"]["counter"] = np.array(w[2:6], dtype=np.uint64)
        st["buffer"] = np.array(w[6:10], dtype=np.uint64)
        st["buffer_pos"] = w[10]
        st["has_uint32"], st["uinteger"] = 0, 0
        bg.state = st
    else:
        bg = np.random.PCG64()
        st = bg.state
        st["state"]["state"] = (w[0] << 64) | w[1]
        st["state"]["inc"] = (w[2] << 64) | w[3]
        st["has_uint32"], st["uinteger"] = 0, 0
        bg.state = st
    return np.random.Generator(bg)


# ---------------------------------------------------------------------------------------
# engine base
# ---------------------------------------------------------------------------------------
PATHS = ("auto", "step", "opaque")


class ManyChainSampler:
    """Common state of the HIP samplers (not part of the reference's API surface).

    Engine options shared by HMCDiag / MALA / DrGhmcDiag (keyword-only, after the reference's own arguments):

    ``chains``, ``chain_id0``   number of chains of a batched device model; global id of the first (Philox key =
                                (seed, global chain id): the same chain at any number of ranks)
    ``path``                    HOW a model's launches are composed; every path gives the same draws bit for bit:
        "auto"    (default) everything the model offers: whole-draw / whole-trajectory / whole-proposal kernels with the
                  density inlined, else one launch per leapfrog step, else the gradient as a separate op
        "step"    no whole-trajectory kernels: one launch per leapfrog step where the model has ``bk_leapfrog_step``
                  (density inlined in the library's step kernel), else the gradient as a separate op
        "opaque"  the gradient ALWAYS a separate device op, whatever the model offers: the model-opaque path BASELINE's
                  56 D bytes per chain-step are counted on (bench.py's headline)
    ``tuning``                  a dict of engine knobs for experiments and bisection (they may also be given as plain
                                keywords); none of them changes a result.  Each class lists its own (``TUNING``).
    ``ops``                     the kernel library object (tests inject a CPU stand-in)
    """

    TUNING: tuple = ()

    @staticmethod
    def _resolve_path(path):
        """path -> (fuse_builtin, fuse_steps): whole-trajectory kernels allowed; one-launch steps allowed."""
        if path not in PATHS:
            raise ValueError(f"path must be one of {PATHS}, got {path!r}")
        return {"auto": (True, True), "step": (False, True), "opaque": (False, False)}[path]

    def _resolve_tuning(self, tuning, knobs):
        """One dict from `tuning=` and loose keywords; unknown names are refused with the list of known ones."""
        out = dict(tuning or {})
        for k, v in knobs.items():
            if k in out and out[k] is not v:
                raise TypeError(f"tuning knob {k!r} given twice")
            out[k] = v
        bad = sorted(set(out) - set(self.TUNING))
        if bad:
            raise TypeError(f"{type(self).__name__}: unknown option(s) {bad}; engine options are chains, chain_id0, path, "
                            f"tuning, ops and the tuning knobs {sorted(self.TUNING)}")
        return out

    def _setup(self, model, metric_diag, init, seed, chains, chain_id0, ops):
        self._model = model
        self._dim = int(model.dims())
        self._batched = getattr(model, "batched", False) is True  # (a Mock's attributes are truthy)
        self._ops = ops if ops is not None else _lib.default_ops()
        dev = self._ops.device
        D = self._dim
        init_t = None
        if init is not None:
            init_t = torch.as_tensor(init)
            if tuple(init_t.shape) == (0,):  # bayes_kit/hmc.py:24-28: an empty init is "absent"
                init_t = None
        if self._batched:
            if chains is None:
                if init_t is not None and init_t.dim() == 2:
                    chains = init_t.shape[0]
                else:
                    raise ValueError("a batched model needs chains=<number of chains> (or a (C, D) init)")
        else:
            if chains not in (None, 1):
                raise ValueError("a single-chain (reference-style) model runs exactly one chain; "
                                 "wrap the density in a batched model to run many")
            chains = 1
        C = self._C = int(chains)
        self._chain_id0 = int(chain_id0)
        self._rng_kind, self._rng_state = make_streams(seed, C, chain_id0, not self._batched, dev)
        # Scratch that lets bk_momentum_refresh put one wavefront (not one lane) on each chain:
        # 5-17x faster while one lane per chain leaves SIMDs empty (C < 65,536), still 1.3x at
        # 65,536 x 1024 (config 3 +1.3 %, config 4 +5 % per draw, same box A/B).
        wave_rng = self._rng_kind == _lib.RNG_PHILOX and D >= 32
        self._rng_work = self._ops.refresh_work(C, D) if wave_rng else None
        self._metric_dev: Optional[torch.Tensor] = None
        self._metric_identity = True
        if metric_diag is not None:
            self._set_metric(metric_diag)
        f64 = dict(dtype=torch.float64, device=dev)
        self._state_pad = self._pick_state_pad(C)
        self._theta_dc = self._new_state()
        if init_t is not None:
            src = init_t.to(dtype=torch.float64)
            if src.dim() == 1:
                if src.shape[0] != D:
                    raise ValueError(f"init has {src.shape[0]} elements, model has {D} dims")
                src = src.reshape(1, D).expand(C, D)
            if tuple(src.shape) != (C, D):
                raise ValueError(f"init must have shape ({C}, {D}) or ({D},), got {tuple(src.shape)}")
            self._theta_dc.copy_(src.t().to(dev))
        else:
            # theta0 = rng.normal(size=D) from the chain's own stream (hmc.py:24-28)
            self._ops.momentum_refresh(self._rng_kind, self._rng_state, None, 0.0, 1.0, self._theta_dc,
                                       None, None, None, self._rng_work)
        self._grad_calls = 0

    # -- metric ----------------------------------------------------------------------------
    def _set_metric(self, m):
        mt = torch.as_tensor(m, dtype=torch.float64).reshape(-1)
        if mt.shape[0] != self._dim:
            raise ValueError(f"metric_diag has {mt.shape[0]} entries, model has {self._dim} dims")
        new = mt.to(self._ops.device).contiguous()
        # a metric of ones (what the reference uses when none is given, hmc.py:22) multiplies by 1.0:
        # an exact identity in IEEE arithmetic, so ALU-bound kernels may be launched without it
        ident = bool((mt == 1.0).all())
        if ident != getattr(self, "_metric_identity", ident):
            self._drop_graphs()  # captured launches chose their kernel variant by this
        self._metric_identity = ident
        if getattr(self, "_metric_dev", None) is not None:
            self._metric_dev.copy_(new)  # in place: a captured hipGraph keeps pointing at this buffer
        else:
            self._metric_dev = new
            self._drop_graphs()  # the captured launches had no metric argument: capture again

    @property
    def _metric(self):
        """Diagonal metric as the reference stores it (hmc.py:22): ones when not given."""
        if self._metric_dev is None:
            return np.ones(self._dim)
        return self._metric_dev.cpu().numpy()

    @_metric.setter
    def _metric(self, m):
        # the reference can only be given a D>1 metric by assigning _metric
        self._set_metric(m)

    # -- views of the state -------------------------------------------------------------------
    @property
    def _theta(self):
        if self._batched:
            return self._theta_dc.t()
        return np.array(self._theta_dc[:, 0].cpu().numpy())

    @property
    def chains(self) -> int:
        return self._C

    def rng_state(self) -> np.ndarray:
        """uint64 [RNG_WORDS, C] snapshot of the per-chain streams (numpy field order)."""
        return self._rng_state.cpu().numpy().view(np.uint64)

    @property
    def _rng(self):
        """A numpy Generator positioned where chain 0's device stream is (read-only view)."""
        return numpy_generator_from_words(self._rng_kind, self.rng_state()[:, 0])

    def __iter__(self):
        return self

    # -- checkpoint / resume -------------------------------------------------------------------
    # The reference keeps a sampler's whole state in plain attributes (_theta, _rng, for
    # DRGHMC _rho and the gradient cache, for MALA the cached logp/grad: hmc.py:21-28,
    # mala.py:31-32, drghmc.py:72-82).  The same here: per-chain tensors plus the RNG table,
    # small and explicit.  A restored sampler continues bit for bit.
    def _state_tensors(self):
        return {"theta": self._theta_dc}

    def _logical_rng(self):
        return self._rng_state

    def _after_load(self):
        pass

    def state_dict(self):
        torch.cuda.synchronize() if self._ops.device.type == "cuda" else None
        sd = {k: v.detach().cpu().clone() for k, v in self._state_tensors().items()}
        sd["rng_state"] = self._logical_rng().detach().cpu().clone()
        sd["meta"] = {"class": type(self).__name__, "chains": self._C, "dims": self._dim,
                      "rng_kind": self._rng_kind, "chain_id0": self._chain_id0,
                      "draws": getattr(self, "_draws", 0), "have_cache": getattr(self, "_have_cache", True),
                      "extra": self._state_extra()}
        return sd

    def _state_extra(self):
        return {}

    def load_state_dict(self, sd):
        meta = sd["meta"]
        if (meta["class"], meta["chains"], meta["dims"]) != (type(self).__name__, self._C, self._dim):
            raise ValueError(f"checkpoint is for {meta['class']} with {meta['chains']} chains x {meta['dims']} dims")
        # the wavefront-per-chain scratch and MALA's normals layout were chosen from the stream kind at
        # construction, and the Philox key IS the global chain id: refuse a checkpoint of other streams
        if meta["rng_kind"] != self._rng_kind:
            raise ValueError(f"checkpoint holds rng kind {meta['rng_kind']}, this sampler was built for {self._rng_kind} "
                             "(Philox = 0, PCG64 = 1): construct it with the same kind of seed")
        if int(meta.get("chain_id0", self._chain_id0)) != self._chain_id0:
            raise ValueError(f"checkpoint is for chains starting at global id {meta['chain_id0']}, "
                             f"this sampler starts at {self._chain_id0}")
        # A previous sample() may have queued the NEXT draw's generator on the side stream; it mutates
        # the RNG table in place and nothing would wait for it once the prefetch bookkeeping is
        # dropped below: let everything in flight finish before the table is overwritten.
        if self._ops.device.type == "cuda":
            torch.cuda.synchronize()
        for k, v in self._state_tensors().items():
            v.copy_(sd[k].to(v.device))
        self._rng_state.copy_(sd["rng_state"].to(self._rng_state.device))
        self._draws = meta["draws"]
        if hasattr(self, "_have_cache"):
            self._have_cache = meta["have_cache"]
        self._load_extra(meta.get("extra", {}))
        self._drop_graphs()  # captured launches may refer to superseded state: capture again
        self._after_load()

    def _load_extra(self, extra):
        pass

    # -- hipGraph replay of a whole draw ----------------------------------------------------------
    # Small problems (e.g. 4096 chains x D=128: 4 MiB arrays) are launch-bound: ~2 L tiny
    # kernels per draw, each costing more host time than device time.  With graph=True the
    # launch sequence of one draw is captured once (after an eager warm-up draw) and replayed
    # as one hipGraph; all per-draw state (RNG table, theta, caches, counters) lives in device
    # memory, so a replay IS the next draw.  Only for samplers whose draw has no host
    # synchronisation (batched models; not the single-chain host-model mode, not DRGHMC).
    GRAPH_AUTO_MAX_ELEMS = 1 << 22  # D*C below which a draw is launch-bound (arrays <= 32 MiB)

    def _init_graph(self, graph, prefetch_rng=None):
        if graph is None and prefetch_rng:
            graph = False  # an explicit request for the side-stream generator: eager launches
        if graph is None:
            # automatic only where capture is known to be safe (the library's own targets: no host
            # synchronisation, no allocation patterns of user code) and where it pays
            graph = (self._batched and hasattr(self._model, "bk_eval") and self._ops.device.type == "cuda"
                     and self._dim * self._C <= self.GRAPH_AUTO_MAX_ELEMS)
        self._use_graph = bool(graph)
        self._graph = None  # the captured draw
        self._graph_warm = 0
        if self._use_graph and not self._batched:
            raise ValueError("graph=True needs a batched device model (the host-model mode synchronises)")

    def _drop_graphs(self):
        self._graph = None
        self._graph_warm = 0

    def _graph_key(self):
        """The host scalars a captured draw bakes into its launches (step sizes, step counts, damping ...:
        plain attributes in the reference, which e.g. a step-size adaptation assigns between draws).  They are
        compared before every replay; a change -- assignment or in-place edit of a list -- captures again."""
        return None

    # A captured draw is a LINEAR graph: samplers that otherwise generate the next draw's randomness on
    # a side stream (prefetch_rng) generate it in line when they replay a graph.  Capturing the side
    # stream as a parallel branch (fork at the start of the draw, join at the end) is both slower
    # (93 vs 79 us per draw at 4096 x 128: the replay pays for the fork/join markers and the branch
    # runs on an internal stream) and unsafe on this ROCm: once such a hipGraphExec is destroyed the
    # runtime's signal thread decrements a counter inside the freed queue object of that internal
    # stream.  tools/heapguard.c pins it down -- a 920-byte block freed in one completion callback of
    # libamdhip64 and written at offset 152 by the next -- and it is what flipped the last bit of
    # unrelated host doubles and tripped glibc's heap checks under tests/soak_samplers.py (several
    # hundred short-lived samplers a second).  Linear graphs, eager two-stream prefetch and plain
    # PyTorch graphs all run clean under the same tool (profiles/r2_heapguard.md).
    def _run_draw(self, draw_fn):
        if not self._use_graph:
            draw_fn()
            return
        if self._graph is None:
            if self._graph_warm < 1:
                draw_fn()  # eager: first-use initialisation, lazy parameter uploads
                self._graph_warm += 1
                return
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            self._graph_scalars = self._graph_key()
            # No garbage collection while the stream is capturing: a collection that happens to run inside the
            # capture may finalise GPU objects of OTHER samplers (graphs, events, streams held in reference
            # cycles), and destroying those is not a capturable operation -- the runtime aborts the process
            # (seen in the full GPU test run, never in a single file: it takes earlier tests' garbage plus a draw
            # that allocates enough Python objects to trigger a collection).  torch.cuda.graph() collects once
            # before the capture begins.
            import gc

            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g):
                    draw_fn()
            finally:
                if gc_was_on:
                    gc.enable()
            self._graph = g
        elif self._graph_key() != self._graph_scalars:
            self._drop_graphs()
            self._graph_warm = 1  # (already warm: capture right away with the new scalars)
            return self._run_draw(draw_fn)
        self._graph.replay()

    def __next__(self):
        return self.sample()

    # -- the RNG side stream ----------------------------------------------------------------------------
    # Events are created ONCE per sampler and re-recorded every draw.  Creating two torch.cuda.Event
    # objects per draw and dropping them while GPU-side waits on them were still pending corrupted the
    # host heap under the randomised soak (glibc "malloc(): invalid size", a chain off by one ulp) --
    # about once per 100,000 short-lived samplers, more often the more events a draw used
    # (tensor.record_stream() made it frequent).  With persistent events, a GPU-side join at the end of
    # every draw and a drained side stream before the sampler's buffers are released, it is gone.
    def _init_side_stream(self):
        dev = self._ops.device
        self._side = torch.cuda.Stream(device=dev)
        self._ev_ready = [torch.cuda.Event(), torch.cuda.Event()]  # main -> side: slot free, RNG table consistent
        self._ev_done = [torch.cuda.Event(), torch.cuda.Event()]   # side -> main: slot filled

    def _join_side_stream(self):
        """Order everything queued later on the main stream after the side stream's pending generator
        (its buffers were allocated on the main stream: without this a sampler dropped right after a draw
        could see them handed to the next allocation while still being written).  A GPU-side wait: free
        when the generator has finished, and the next draw's first kernel waits for the same event anyway."""
        ev = getattr(self, "_pf_event", None)
        if ev is not None and getattr(self, "_prefetch", False):
            torch.cuda.current_stream().wait_event(ev)

    def __del__(self):
        try:
            side = getattr(self, "_side", None)
            if side is not None:
                side.synchronize()  # nothing of this sampler is in flight when its events and buffers go
        except Exception:  # interpreter shutdown
            pass

    # -- model bridge ----------------------------------------------------------------------------
    def _eval_grad(self, theta_dc, grad_out, logp_out):
        """Gradient (and, if logp_out is given, log density) at theta_dc [D, n].

        Returns the tensor that holds the gradient as a logical [D, n] array: ``grad_out``
        itself for built-in targets and host models, or the model's own output (any
        strides, not copied) for generic batched PyTorch models.
        """
        self._grad_calls += 1
        m = self._model
        if hasattr(m, "bk_eval"):
            m.bk_eval(theta_dc, grad_out, logp_out)
            return grad_out
        if self._batched:
            if logp_out is None and hasattr(m, "bk_gradient"):
                lp, g = None, m.bk_gradient(theta_dc.t())   # (a model that can give the gradient without the log density)
            else:
                lp, g = m.log_density_gradient(theta_dc.t())
            if g.dtype != torch.float64 or g.device != theta_dc.device:
                g = g.to(device=theta_dc.device, dtype=torch.float64)  # the kernels read raw fp64 pointers
            if tuple(g.shape) != (theta_dc.shape[1], theta_dc.shape[0]):
                raise ValueError(f"log_density_gradient must return a ({theta_dc.shape[1]}, {theta_dc.shape[0]}) "
                                 f"gradient, got {tuple(g.shape)}")
            if logp_out is not None:
                logp_out.copy_(lp.reshape(-1))
            return g.t()
        th = np.array(theta_dc[:, 0].cpu().numpy())
        lp, g = m.log_density_gradient(th)
        g = np.array(np.broadcast_to(np.asarray(g, dtype=np.float64), (self._dim,)))  # array-likes allowed (mala.py:32)
        grad_out[:, 0].copy_(torch.from_numpy(g))
        if logp_out is not None:
            logp_out.fill_(float(lp))
        return grad_out

    def _eval_logp(self, theta_dc, logp_out):
        m = self._model
        if hasattr(m, "bk_eval"):
            m.bk_eval(theta_dc, None, logp_out)
        elif self._batched:
            logp_out.copy_(m.log_density(theta_dc.t()))
        else:
            lp = m.log_density(np.array(theta_dc[:, 0].cpu().numpy()))
            try:
                logp_out.fill_(float(lp))
            except (TypeError, ValueError):
                # the reference stores whatever the model returns (metropolis.py:99) and only
                # fails when it is compared; a non-numeric value becomes NaN (never accepted)
                logp_out.fill_(float("nan"))

    def _materialize(self, g, grad_out):
        """Make sure the gradient lives in grad_out ([D, n], chain-contiguous)."""
        if g is grad_out or g.data_ptr() == grad_out.data_ptr():
            return grad_out
        self._ops.relayout(g, grad_out)
        return grad_out

    # -- which allocation plays which role -------------------------------------------------------
    # The hot loops stream several arrays at equal offsets.  How fast that runs depends on where the
    # driver placed them RELATIVE to each other: the same kick+drift launch takes 414-484 us on
    # different triples of identically sized allocations (6.5 -> 5.6 TB/s; tools/attic/placement_probe.py arrays,
    # examples/c_host/placement_probe.c), while each array alone streams at the same rate.  A process
    # cannot choose physical placement, but it can choose which of its allocations plays which role:
    # a few assignments of the scratch arrays (plus spare ones, freed afterwards) are timed with the
    # real kernels and the fastest is kept.  Scratch contents are irrelevant at that point (every array
    # is written before it is read); results do not depend on the assignment.
    TUNE_PLACEMENT_MIN_BYTES = 128 << 20
    TUNE_PLACEMENT_TRIALS = 40
    TUNE_PLACEMENT_SPARES = 5

    def _wants_placement_tuning(self, tune_placement):
        if tune_placement is not None:
            return bool(tune_placement)
        return (self._batched and self._ops.device.type == "cuda" and not self._use_graph
                and self._dim * self._C * 8 >= self.TUNE_PLACEMENT_MIN_BYTES)

    def _tune_roles(self, arrays, cost):
        """arrays: interchangeable [D, C] scratch tensors, in role order.  cost(list in role order) ->
        milliseconds.  Returns (the best assignment as a list in role order, report dict)."""
        import random

        pool = list(arrays)
        try:
            # spare candidates come straight from the driver (hipMalloc), not from PyTorch's caching allocator: the ones
            # that are not chosen go back to the driver when they are dropped, with no torch.cuda.empty_cache() -- a
            # sampler's constructor must not flush the user's allocator
            pool += [self._new_state(raw=True) for _ in range(self.TUNE_PLACEMENT_SPARES)]
        except (torch.cuda.OutOfMemoryError, _lib.BkHipError):
            pool = list(arrays)  # no room for spare candidates: permute what there is
        for a in pool:
            a.zero_()  # timing on defined values
        rnd = random.Random(0)
        ids = list(range(len(pool)))
        candidates = [ids[:]]  # the allocation order itself
        while len(candidates) < self.TUNE_PLACEMENT_TRIALS:
            perm = ids[:]
            rnd.shuffle(perm)
            candidates.append(perm)
        best, best_ms, first_ms = None, float("inf"), None
        for perm in candidates:
            ms = cost([pool[i] for i in perm[:len(arrays)]])
            first_ms = ms if first_ms is None else first_ms
            if ms < best_ms:
                best, best_ms = perm, ms
        chosen = [pool[i] for i in best[:len(arrays)]]
        torch.cuda.synchronize()  # (the timing launches are done before any spare is handed back)
        del pool  # unchosen spares: hipFree; displaced originals: back into PyTorch's cache, like any dropped tensor
        return chosen, {"ms_as_allocated": first_ms, "ms_chosen": best_ms, "assignments_tried": len(candidates)}

    @staticmethod
    def _time_ms(fn, warm=2, reps=6):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    _out = None
    _state_pad = 0
    # Columns of padding in the row pitch of a sampler's [D, C] arrays when C * 8 bytes is a multiple of 4 KiB
    # (0 = none).  Kernels that walk DOWN the rows of a few chains (per-chain reductions, the one-pass MALA step) then
    # find every row of those chains on the same memory channels; 1152 bytes off the power-of-two pitch spreads them.
    # Measured per sampler (the streaming HMC kernels do not care: 39.1 ms per draw with or without), so a class
    # attribute; BK_STATE_PAD=<columns> overrides it for experiments.
    STATE_PAD_COLUMNS = 0

    def _pick_state_pad(self, C):
        env = os.environ.get("BK_STATE_PAD")
        if env is not None:
            return max(0, int(env))
        return self.STATE_PAD_COLUMNS if (self._batched and C >= 4096 and (C * 8) % 4096 == 0) else 0

    def _new_state(self, raw=False):
        """A [D, C] state array (rows `_state_pad` columns further apart than C).  raw: memory of its own from the
        driver instead of PyTorch's caching allocator (_lib.RawDeviceArray)."""
        shape = (self._dim, self._C + self._state_pad)
        if raw and self._ops.device.type == "cuda":
            t = _lib.RawDeviceArray(shape).tensor()
        else:
            t = torch.empty(shape, dtype=torch.float64, device=self._ops.device)
        return t[:, :self._C] if self._state_pad else t

    def _select(self, mask, th, thp, g=None, gp=None):
        """theta (and its cached gradient) <- proposal on the accepted chains; in the same pass
        the array sample() returns is written (a fresh tensor per draw: the reference rebinds
        ``_theta`` instead of mutating it, so returned draws must stay valid)."""
        self._out = None
        if self._batched and not self._use_graph:
            self._out = self._new_state()
        self._ops.select_columns(mask, th, thp, g, gp, self._out)

    def _draw_out(self, theta_dc, logp):
        """What sample() hands back: stable copies, shaped like the reference's return."""
        if self._batched:
            out, self._out = self._out, None
            if out is None:
                out = torch.empty(tuple(theta_dc.shape), dtype=theta_dc.dtype, device=theta_dc.device)
                out.copy_(theta_dc)  # (a dense [D, C] copy also when the state's rows are padded)
            return out.t(), logp.clone()
        return np.array(theta_dc[:, 0].cpu().numpy()), np.float64(logp[0].item())
