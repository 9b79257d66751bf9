"""Randomised differential soak: many-chain HIP samplers vs the NumPy oracle (one object per chain).

Random algorithm (HMC / MALA / DRGHMC / Metropolis), target (iso / diag Gaussian / AR(1)), model provider
(library target, PyTorch autograd, user code returning a row-major or strided gradient, compiled plugin),
dims (1..70: both generator kernels), chains (odd and even: scalar and 16-byte kernels), step sizes, trajectory lengths,
metric on/off, fused / step-by-step, hipGraph on/off, RNG prefetch on/off.  For a few watched chains
theta must be bit-identical to the oracle at every draw and the stream state equal at the end.
SECONDS env var = duration (default 60); SEED = rng seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
import bayes_kit_amd as bk
from bayes_kit_amd.metropolis import ChainRng
from oracle import models as om
from oracle import samplers as osamp
from tests import provider_parity as pp
from tests.host_models import Ar1


if os.environ.get("SOAK_NO_FUSED_DRAW"):
    bk.HMCDiag.ENABLE_FUSED_DRAW = False
if os.environ.get("SOAK_NO_FUSED_ZT"):
    bk.HMCDiag.ENABLE_FUSED_ZT = False
MALA_KW = dict(two_pass=False) if os.environ.get("SOAK_NO_TWO_PASS") else {}


NO_GRAPH = bool(os.environ.get("SOAK_NO_GRAPH"))
NO_PREFETCH = bool(os.environ.get("SOAK_NO_PREFETCH"))


def _forced(name, drawn):
    v = os.environ.get(name)
    return drawn if v is None else v == "1"


class Sentinels:
    """Zero-filled host blocks scattered through the malloc arena: any word that stops being zero was
    written by somebody else (SOAK_SENTINELS=1)."""

    def __init__(self, n=4000, words=64):
        blocks = [np.zeros(words, dtype=np.int64) for _ in range(2 * n)]
        self.keep = blocks[::2]  # every other block is freed again: holes for the samplers' own allocations
        del blocks

    def check(self):
        for i, b in enumerate(self.keep):
            if b.any():
                j = np.nonzero(b)[0]
                raise AssertionError(("sentinel", i, hex(b.ctypes.data), j.tolist(), [hex(int(v) & (2**64 - 1)) for v in b[j]]))


PROVIDERS_SEEN = {}
HISTORY = []  # the last few configurations: a corrupted sampler is usually the victim of an earlier one


def funnel_paths(rng):
    """DRGHMC on the built-in funnel (not bit-comparable with the oracle: the kernel sums a chain's coordinates in its
    own canonical order) -- the device-side lane counts with every launch fusion (a proposal's first ghost, ghost
    links, the scatter riding on the next stage, refresh + start of the draw in one launch; replayed as a hipGraph
    or not) against launches sized by host reads with every ghost a launch of its own: bit for bit, with the
    trajectories run, the momenta and the stream positions."""
    D = int(rng.choice([2, 3, 11, 17, 18, 33, 34, 50, 65, 101, 129]))
    C = int(rng.choice([1, 3, 64, 65, 130, 700, 2100, 5000, 13000]))
    K = int(rng.integers(1, 5))
    eps = float(rng.uniform(0.05, 0.6))
    sizes = [eps / (3 ** k) for k in range(K)]
    counts = [int(rng.integers(1, 5)) * (2 ** k) for k in range(K)]
    damp = float(rng.uniform(0.05, 1.0))
    pr = bool(rng.integers(0, 2))
    metric = np.linspace(0.7, 1.4, D) if rng.random() < 0.4 else None
    seed = int(rng.integers(1, 2**40))
    N = int(rng.integers(2, 9))
    graph = bool(rng.integers(0, 2)) and not NO_GRAPH
    fuse = bool(rng.integers(0, 4))
    regrad = bool(rng.integers(0, 4))   # (round 6) the one-launch path without a gradient cache, or with
    desc = dict(alg="drfunnel", D=D, C=C, K=K, sizes=sizes, counts=counts, damp=damp, prob_retry=pr, metric=metric is not None,
                seed=seed, N=N, graph=graph, fuse_first_ghost=fuse, recompute_gradient=regrad)
    mk = lambda **kw: bk.DrGhmcDiag(bk.Funnel(D), K, sizes, counts, damp, metric_diag=metric, chains=C, seed=seed,  # noqa: E731
                                    prob_retry=pr, **kw)
    a = mk(device_counts=False)
    b = mk(device_counts=True, graph=graph, fuse_first_ghost=fuse, recompute_gradient=regrad)
    # round 4: the gradient as a separate COUNTED op per leapfrog step (built-in op, or the user plugin), lane counts on
    # the device -- against the same host-sized path (step by step, so that the joint log densities agree bit for bit too)
    # round 5: "builtin" runs {gradient, kick, drift} as ONE launch per step (bk_leapfrog_step), "builtin_op" keeps the gradient a
    # separate op; "source" / "source_op": the funnel compiled from lanes-form source on the counted path, with / without the
    # one-launch step; "source_fused": the same source through the one-launch proposal kernel (against the built-in's)
    # "chain" / "chain_step" / "chain_op": the funnel as a PER-CHAIN source (class-order sums: bk.Funnel's bits) with one launch per
    # trajectory (bk_leapfrog_trajectory, D <= 128), one per leapfrog step, or the gradient a separate op; counted or host-sized
    opaque = str(rng.choice(["none", "builtin", "builtin_op", "plugin", "source", "source_op", "source_fused", "chain", "chain_step",
                             "chain_op"]))
    desc["opaque"] = opaque
    o = a2 = None
    if opaque != "none":
        a2 = mk(device_counts=False, path=("step", "opaque")[int(rng.integers(0, 2))])
        if opaque.startswith("source"):
            src = bk.CTarget.from_source(FUNNEL_LANES_SRC, D, form="lanes", head=1)
            kw = dict(source=dict(path="step"), source_op=dict(path="opaque"), source_fused=dict())[opaque]
            o = bk.DrGhmcDiag(src, K, sizes, counts, damp, metric_diag=metric, chains=C, seed=seed, prob_retry=pr,
                              device_counts=True, graph=graph, **kw)
            if opaque == "source_fused":
                a2 = mk(device_counts=True, graph=graph)   # (the fused kernels sum the kinetic energy in their lanes' order)
        elif opaque.startswith("chain"):
            src = bk.CTarget.from_source(FUNNEL_CHAIN_CLASS_ORDER_SRC, D, form="chain")
            kw = dict(chain=dict(), chain_step=dict(path="step"), chain_op=dict(path="opaque"))[opaque]
            dc = bool(rng.integers(0, 2))
            desc["chain_device_counts"] = dc
            o = bk.DrGhmcDiag(src, K, sizes, counts, damp, metric_diag=metric, chains=C, seed=seed, prob_retry=pr,
                              device_counts=dc, graph=graph and dc, **kw)
            assert o._traj_hook == (opaque == "chain" and D <= 128), desc
        elif opaque == "builtin_op":
            o = mk(device_counts=True, graph=graph, path="opaque")
        elif opaque == "plugin":
            import os as _os

            lib = _os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so")
            o = bk.DrGhmcDiag(bk.CTarget(lib, "funnel_target", D, counted_symbol="funnel_target_n"), K, sizes, counts, damp,
                              metric_diag=metric, chains=C, seed=seed, prob_retry=pr, device_counts=True, graph=graph)
        else:
            o = mk(device_counts=True, graph=graph, path="step")
    for n in range(N):
        ta, la = a.sample()
        tb, lb = b.sample()
        same = torch.equal(ta, tb) or (torch.isnan(ta) == torch.isnan(tb)).all() and torch.equal(ta.nan_to_num(), tb.nan_to_num())
        assert same and torch.equal(la.nan_to_num(), lb.nan_to_num()), ("theta", desc, n)
        assert a.last_stage_lanes == b.last_stage_lanes and a.last_lane_steps == b.last_lane_steps, ("lanes", desc, n)
        if o is not None:
            t2, l2 = a2.sample()
            to, lo = o.sample()
            assert torch.equal(ta.nan_to_num(), to.nan_to_num()) and torch.equal(t2.nan_to_num(), to.nan_to_num()), ("opaque theta", desc, n)
            assert torch.equal(l2.nan_to_num(), lo.nan_to_num()), ("opaque logp", desc, n)
            assert a.last_stage_lanes == o.last_stage_lanes, ("opaque lanes", desc, n)
    assert torch.equal(a._rho.nan_to_num(), b._rho.nan_to_num()), ("rho", desc)
    assert np.array_equal(a.rng_state(), b.rng_state()), ("stream", desc)
    if o is not None:
        assert torch.equal(a._rho.nan_to_num(), o._rho.nan_to_num()), ("opaque rho", desc)
        assert np.array_equal(a.rng_state(), o.rng_state()), ("opaque stream", desc)
    # round 6: advance(n) -- graphs of several consecutive draws, an attached moments update riding on the next draw's
    # generator launch -- against n advance() calls: state, momenta, streams, moments and series
    if graph and rng.random() < 0.5:
        M = int(rng.integers(3, 30))
        desc["advance_n"] = M
        mom_b, mom_a = bk.RunningMoments(D, C), bk.RunningMoments(D, C)
        rec_b, rec_a = bk.DrawRecorder([0, D - 1], M, C), bk.DrawRecorder([0, D - 1], M, C)
        b.attach(moments=mom_b, recorder=rec_b)
        b.advance(M)
        for _ in range(M):
            ta, la = a.sample()
            mom_a.update(ta)
            rec_a.record(ta, la)
        assert torch.equal(a._theta_dc.nan_to_num(), b._theta_dc.nan_to_num()) and torch.equal(a._rho.nan_to_num(), b._rho.nan_to_num()), ("advance(n) state", desc)
        assert np.array_equal(a.rng_state(), b.rng_state()), ("advance(n) stream", desc)
        assert mom_a.n == mom_b.n == M and torch.equal(mom_a.mean.nan_to_num(), mom_b.mean.nan_to_num()) \
            and torch.equal(mom_a.m2.nan_to_num(), mom_b.m2.nan_to_num()), ("advance(n) moments", desc)
        assert torch.equal(rec_a.series.nan_to_num(), rec_b.series.nan_to_num()), ("advance(n) series", desc)
        b.detach()
    return "drfunnel"


FUNNEL_CHAIN_CLASS_ORDER_SRC = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* /*params*/) {
  const double v = th[0];
  double cs[16];
  for (int c = 0; c < 16; ++c) {
    double a = 0.0;
    for (i64 d = 1 + c; d < D; d += 16) { const double x = th[d]; a = a + x * x; }
    cs[c] = a;
  }
  double q[4];
  for (int k = 0; k < 4; ++k) q[k] = ((cs[k] + cs[k + 4]) + cs[k + 8]) + cs[k + 12];
  const double s = ((q[0] + q[1]) + q[2]) + q[3];
  const double ev = bk_exp(-v), hn = 0.5 * (double)(D - 1), he = 0.5 * ev;
  if (g.wanted()) {
    g.set(0, ((-v / 9.0) - hn) + he * s);
    for (i64 d = 1; d < D; ++d) g.set(d, -(ev * th[d]));
  }
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}
"""

FUNNEL_LANES_SRC = """
template <class L>
__device__ double bk_lanes_density(L& c, const double* /*params*/) {
  const double v = c.head(0);
  const double s = c.sum([](double x, i64) { return x * x; });
  const double ev = bk_exp(-v);
  const double hn = 0.5 * (double)(c.dims() - 1);
  const double he = 0.5 * ev;
  c.grad_head(0, ((-v / 9.0) - hn) + he * s);
  c.grad([ev](double x, i64) { return -(ev * x); });
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}
"""


def hmc_funnel(rng):
    """Round 5: plain HMC on the funnel through the lane-spread kernels -- whole trajectory one launch (built-in or from source),
    one launch per leapfrog step, gradient a separate op -- same draws bit for bit, whatever the step size does to the
    trajectory (overflow to inf / NaN included); the watched chains against the oracle while the flow is still tame."""
    D = int(rng.choice([2, 3, 11, 17, 33, 50, 101, 129, 200]))
    C = int(rng.choice([1, 3, 64, 65, 700, 2100, 13000]))
    eps = float(rng.uniform(0.02, 0.5))
    L = int(rng.integers(1, 12))
    metric = np.linspace(0.7, 1.4, D) if rng.random() < 0.4 else None
    seed = int(rng.integers(1, 2**40))
    N = int(rng.integers(2, 7))
    graph = None if rng.integers(0, 2) else (bool(rng.integers(0, 2)) and not NO_GRAPH)
    desc = dict(alg="hmcfunnel", D=D, C=C, eps=eps, L=L, metric=metric is not None, seed=seed, N=N, graph=graph)
    # (the funnel as a per-chain source too: one launch per trajectory for D <= 128 -- bk_leapfrog_trajectory --, per step, separate op)
    kind = int(rng.integers(0, 3))
    desc["model"] = ("builtin", "lanes source", "chain source")[kind]
    model = (lambda: bk.Funnel(D), lambda: bk.CTarget.from_source(FUNNEL_LANES_SRC, D, form="lanes", head=1),
             lambda: bk.CTarget.from_source(FUNNEL_CHAIN_CLASS_ORDER_SRC, D, form="chain"))[kind]
    mk = lambda **kw: bk.HMCDiag(model(), eps, L, metric_diag=metric, chains=C, seed=seed, graph=graph, **kw)  # noqa: E731
    f, h, s_ = mk(), mk(path="step"), mk(path="opaque")
    for n in range(N):
        tf, lf = f.sample()
        th_, lh = h.sample()
        ts, ls = s_.sample()
        assert torch.equal(th_.nan_to_num(), ts.nan_to_num()) and torch.equal(lh.nan_to_num(), ls.nan_to_num()), ("step hook", desc, n)
        assert torch.equal(tf.nan_to_num(), ts.nan_to_num()), ("one-launch trajectory", desc, n)
    assert np.array_equal(f.rng_state(), s_.rng_state()) and np.array_equal(h.rng_state(), s_.rng_state()), ("stream", desc)
    return "hmcfunnel"


def one(rng, it):
    alg = rng.choice(os.environ["ALGS"].split(",")) if os.environ.get("ALGS") else rng.choice(["hmc", "mala", "drghmc", "metropolis", "drfunnel", "hmcfunnel"])
    if alg == "hmcfunnel":
        return hmc_funnel(rng)
    if alg == "drfunnel":
        return funnel_paths(rng)
    if alg == "torchgraph":
        # control: a hipGraph of two plain PyTorch kernels, captured and replayed the way a sampler's draw is
        x = torch.zeros(int(rng.choice([3, 64, 500])), dtype=torch.float64, device="cuda")
        x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            x.add_(1.0)
            x.mul_(1.0)
        for _ in range(int(rng.integers(2, 10))):
            g.replay()
        assert float(x[0].item()) >= 3.0
        return "torchgraph"
    if alg == "torchfork":
        # control: the same with a parallel branch (fork onto a side stream, join), PyTorch kernels only,
        # shaped like the samplers' former forked capture: a device-to-device copy and a few kernels on
        # the branch, two graphs sharing the side stream, replayed alternately
        n = int(rng.choice([3, 64, 500]))
        f64 = dict(dtype=torch.float64, device="cuda")
        x, y, z = torch.zeros(n, **f64), torch.zeros((11, n), **f64), torch.zeros((11, n), **f64)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        graphs = []
        for _ in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    z.copy_(y)
                    y.add_(1.0)
                    y.mul_(1.0)
                x.add_(1.0)
                x.mul_(1.0)
                x.add_(0.0)
                main.wait_stream(side)
            graphs.append(g)
        reps = int(rng.integers(2, 10))
        for i in range(reps):
            graphs[i % 2].replay()
        got = x.cpu().numpy()
        assert got[0] == reps and float(y[0, 0].item()) == reps and float(z[0, 0].item()) == reps - 1
        return "torchfork"
    D = int(rng.choice([1, 2, 7, 31, 32, 33, 40, 64, 70]))
    C = int(rng.choice([1, 2, 3, 63, 64, 65, 130, 257, 500]))
    N = int(rng.integers(3, 12))
    seed = int(rng.integers(1, 2**40))
    iso = rng.random() < 0.4
    lam = None if iso else np.logspace(0, rng.uniform(0.2, 1.0), D)
    # who supplies the gradient: the library's own target, autograd of user torch code,
    # user code with a gradient layout of its own, or a compiled plugin behind bk_target_fn
    prov = str(rng.choice(os.environ["PROVIDERS"].split(","))) if os.environ.get("PROVIDERS") else \
        str(rng.choice(["builtin", "builtin", "builtin", "torch", "row", "strided", "plugin"]))
    lam_eff = np.ones(D) if iso else lam
    if prov == "builtin":
        tgt = bk.IsoGaussian(D) if iso else bk.DiagGaussian(lam)
    elif prov == "torch":
        tgt = pp.torch_diag_gaussian(lam_eff, "cuda")
    elif prov in ("row", "strided"):
        tgt = pp.RowMajorDiag(lam_eff, "cuda", prov)
    else:
        a_ar, s2_ar = float(rng.uniform(-0.8, 0.8)), float(rng.uniform(0.5, 1.5))
        tgt = pp.ar1_plugin(D, a_ar, s2_ar)
    if prov == "plugin":
        otgt = lambda: Ar1(D, a_ar, s2_ar)
    else:
        otgt = (lambda: om.IsoGaussian(D)) if iso else (lambda: om.DiagGaussian(lam))
    metric = np.linspace(0.8, 1.2, D) if (rng.random() < 0.4 and alg in ("hmc", "drghmc")) else None
    eps = float(rng.uniform(0.02, 0.3))
    desc = dict(alg=str(alg), D=D, C=C, N=N, seed=seed, iso=bool(iso), metric=metric is not None, eps=eps, provider=prov)
    PROVIDERS_SEEN[prov] = PROVIDERS_SEEN.get(prov, 0) + 1
    HISTORY.append(desc)
    del HISTORY[:-8]
    if alg == "hmc":
        L = int(rng.integers(0, 7))
        kw = dict(path=("step", "auto")[int(rng.integers(0, 2))], graph=bool(rng.integers(0, 2)) and not NO_GRAPH,
                  prefetch_rng=bool(rng.integers(0, 2)) and not NO_PREFETCH)
        desc.update(L=L, **kw)
        s = bk.HMCDiag(tgt, eps, L, metric_diag=metric, chains=C, seed=seed, **kw)
        mk = lambda sd: osamp.HMCDiag(otgt(), eps, L, metric_diag=metric, seed=sd)
    elif alg == "mala":
        kw = dict(graph=_forced("SOAK_MALA_GRAPH", bool(rng.integers(0, 2)) and not NO_GRAPH),
                  prefetch_rng=_forced("SOAK_MALA_PREFETCH", bool(rng.integers(0, 2)) and not NO_PREFETCH))
        desc.update(**kw)
        s = bk.MALA(tgt, eps * 0.3, chains=C, seed=seed, **kw, **MALA_KW)
        mk = lambda sd: osamp.MALA(otgt(), eps * 0.3, seed=sd)
    elif alg == "drghmc":
        K = int(rng.integers(1, 4))
        sizes = [float(eps * 2.0 / (2 ** k)) for k in range(K)]
        counts = [int(rng.integers(1, 4)) * (2 ** k) for k in range(K)]
        damp = float(rng.uniform(0.05, 1.0))
        pr = bool(rng.integers(0, 2))
        desc.update(K=K, sizes=sizes, counts=counts, damp=damp, prob_retry=pr)
        s = bk.DrGhmcDiag(tgt, K, sizes, counts, damp, metric_diag=metric, chains=C, seed=seed, prob_retry=pr)
        mk = lambda sd: osamp.DrGhmcDiag(otgt(), K, sizes, counts, damp, metric_diag=metric, seed=sd, prob_retry=pr)
    else:
        scale = float(rng.uniform(0.2, 1.5))
        pseed = seed + 17
        prop = ChainRng(pseed, C)
        desc.update(scale=scale)
        s = bk.Metropolis(tgt, lambda Th: prop.normal(Th, scale), chains=C, seed=seed)

        def mk(sd, pseed=pseed, scale=scale):
            c = int(sd.state["state"]["key"][1])
            g = np.random.Generator(np.random.Philox(key=[pseed, c]))
            return osamp.Metropolis(otgt(), lambda th: g.normal(loc=th, scale=scale), seed=sd)
    watch = sorted(set([0, C // 2, C - 1]))
    got = []
    for n in range(N):
        th, _ = s.sample()
        got.append(th[watch].cpu().numpy())
    state = s.rng_state()
    if os.environ.get("SOAK_SYNC_BEFORE_DROP"):
        torch.cuda.synchronize()
    for j, c in enumerate(watch):
        o = mk(np.random.Philox(key=[seed, c]))
        for n in range(N):
            oth, _ = o.sample()
            if not np.array_equal(oth, got[n][j]):
                bad = np.nonzero(oth != got[n][j])[0]
                raise AssertionError(("theta", desc, dict(chain=c, draw=n, dims_differing=bad[:8].tolist(),
                                                          n_differing=int(bad.size),
                                                          max_abs=float(np.abs(oth - got[n][j]).max()),
                                                          lam=None if lam is None else [float(lam[0]), float(lam[-1])])))
        st = o._rng.bit_generator.state
        assert [int(v) for v in st["state"]["counter"]] == [int(v) for v in state[2:6, c]], ("counter", desc, c)
        assert int(st["buffer_pos"]) == int(state[10, c]), ("pos", desc, c)
    return alg


if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
    budget = float(os.environ.get("SECONDS", 60))
    t0, counts = time.time(), {}
    sent = Sentinels() if os.environ.get("SOAK_SENTINELS") else None
    guard = os.environ.get("SOAK_HEAPGUARD")  # tools/heapguard.c, which must also be in LD_PRELOAD
    while time.time() - t0 < budget:
        try:
            if guard and sum(counts.values()) == 40:
                import ctypes, faulthandler
                faulthandler.enable()
                ctypes.CDLL(guard).heapguard_enable()
            a = one(rng, sum(counts.values()))
            if sent is not None:
                sent.check()
        except BaseException:
            print("history (oldest first):", *HISTORY, sep="\n  ", flush=True)
            raise
        counts[a] = counts.get(a, 0) + 1
    print("soak ok:", sum(counts.values()), "random sampler configurations", counts, "providers", PROVIDERS_SEEN,
          f"in {time.time()-t0:.0f} s")
