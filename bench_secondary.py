"""The secondary workloads of bench.py (`secondary` in its JSON line; `python bench.py --only NAME` runs one alone): the
other BASELINE.json configs and the model providers, each a few draws with its own bound and bytes / flop model.  Never
part of `value`; a failure here never costs the headline line."""
import os
import time

from bench_config import (C_CFG3, D_CFG3, EPS_CFG3, FP64_MFMA_PEAK_TFLOPS, FP64_VECTOR_PEAK_TFLOPS, HBM_PEAK_GBPS,
                           # noqa: F401
                          L_CFG3, SEED_CFG3)

ROOT = os.path.dirname(os.path.abspath(__file__))


def bench_cfg2(ctx, steps=200, warmup=6, chains=4096):
    """configs[1]: iso-Gaussian D=128, HMC L=32, 4096 chains: arrays of 4 MiB, cache-resident and
    launch/latency-bound.  Model-opaque (separate gradient op, hipGraph replay) and fused."""
    import torch

    import bayes_kit_amd as bk

    C, D, L = chains, 128, 32
    res = {"workload": "BASELINE.json configs[1]: iso-Gaussian D=128, HMC L=32 eps=0.05, 4096 chains",
           "bound": "launch/latency",
           "note": "4 MiB arrays live in L2 / Infinity Cache: an HBM fraction is not meaningful; us per leapfrog step "
                   "of all chains is the figure"}
    # model_opaque: the gradient a separate op per leapfrog step; one_launch_per_step: {gradient, kick, drift} one
    # launch (bk_leapfrog_step_gaussian); fused_builtin: the whole draw in registers
    for name, path in (("model_opaque", "opaque"), ("one_launch_per_step", "step"), ("fused_builtin", "auto")):
        s = bk.HMCDiag(bk.IsoGaussian(D), 0.05, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=20240,
                       chains=C, chain_id0=ctx.rank * C, path=path)
        for _ in range(warmup):
            s.sample()
        el = ctx.timed_loop(s.sample, steps)
        res[name] = {"us_per_leapfrog_step": 1e6 * el / steps / L, "ms_per_draw": 1e3 * el / steps,
                     "steps_per_sec": C * ctx.world * L * steps / el, "accept_rate": s.accept_rate(),
                     "hipgraph": bool(s._use_graph)}
        del s
    # The same target as five lines of HIP C++ handed to CTarget.from_source: the generated translation unit
    # instantiates the library's whole-draw kernel (csrc/bk_elementwise.hpp) with the user's bk_term inlined -- the path
    # the built-in takes, for ANY separable density.  Reported separately; never priced on the 56*D model.
    try:
        t0 = time.perf_counter()
        model = bk.CTarget.from_source(ISO_TERM_SRC, D)
        build_s = time.perf_counter() - t0
        ref = bk.HMCDiag(bk.IsoGaussian(D), 0.05, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=20240,
                         chains=C,
                         chain_id0=ctx.rank * C, path="step")
        s = bk.HMCDiag(model, 0.05, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=20240, chains=C,
                       chain_id0=ctx.rank * C)
        for _ in range(warmup):
            s.sample()
            ref.sample()
        same = bool(torch.equal(s._theta_dc, ref._theta_dc) and torch.equal(s._rng_state, ref._rng_state))
        el = ctx.timed_loop(s.sample, steps)
        res["compiled_source_fused"] = {
            "what": "CTarget.from_source(<bk_term of the iso Gaussian>): whole draw (trajectory + "
                    "energies + accept) in the "
                    "library's register-resident kernel with the compiled term inlined",
            "us_per_leapfrog_step": 1e6 * el / steps / L, "ms_per_draw": 1e3 * el / steps,
            "steps_per_sec": C * ctx.world * L * steps / el, "accept_rate": s.accept_rate(),
            "hipgraph": bool(s._use_graph),
            "fused_draw": bool(s._fused_draw), "identical_to_step_by_step_builtin": same,
            "construction_s_incl_hipcc_or_cache": build_s}
        del s, ref
    except Exception as e:  # context only
        res["compiled_source_fused"] = {"error": repr(e)}
    return res


ISO_TERM_SRC = """
__device__ __forceinline__ void bk_term(double th, i64 d, const double* /*params*/, double& term, double& grad) {
  term = -0.5 * (th * th);
  grad = -th;
}
"""

FUNNEL_LANES_SRC = """
// Neal's funnel for the lane-spread form (head = 1: v = theta_0 is held by every lane of the chain)
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

FUNNEL_CHAIN_SRC = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* /*params*/) {
  const double v = th[0];
  double s = 0.0;
  for (i64 d = 1; d < D; ++d) { const double x = th[d]; s = s + x * x; }
  const double ev = bk_exp(-v), hn = 0.5 * (double)(D - 1), he = 0.5 * ev;
  if (g.wanted()) {
    g.set(0, ((-v / 9.0) - hn) + he * s);
    for (i64 d = 1; d < D; ++d) g.set(d, -(ev * th[d]));
  }
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}
"""


def bench_cfg4(ctx, draws=40, warmup=3, chains=32768, full_rhat=False, spec_length=True, extras=True):
    """configs[3]: Neal's funnel D=101, DRGHMC K=3, 32,768 chains PER RANK (chain ids rank*C ..), R-hat over
    ALL ranks' chains and the summed ESS through the process group (bayes_kit/rhat.py:163-171: the
    cross-chain reduction north_star assigns to RCCL)."""
    import torch

    import bayes_kit_amd as bk

    C, D = chains, 101
    s = bk.DrGhmcDiag(bk.Funnel(D), 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=C, chain_id0=ctx.rank * C,
                      seed=20242)
    mom = bk.RunningMoments(D, C)
    rec = bk.DrawRecorder([0, 1, D - 1], draws, C)
    for _ in range(warmup):
        s.sample()
    # The Welford moments of all 101 dims and the tracked series (3 dims + joint log density) are fed from
    # INSIDE every draw (DrGhmcDiag.attach): their two launches are part of the draw's replayed hipGraph and read
    # the draw index from the sampler's device-side counter; advance() is a draw without returned copies.
    s.attach(moments=mom, recorder=rec)
    state = {"lane_steps": 0}
    on_device = hasattr(s, "lane_steps_total") and s._dev_counts  # counted on the device: no host read per draw
    base = float(s.lane_steps_total.item()) if on_device else 0.0

    def one():
        s.advance()
        if not on_device:
            state["lane_steps"] += s.last_lane_steps

    el = ctx.timed_loop(one, draws)
    lane_steps = float(s.lane_steps_total.item()) - base if on_device else state["lane_steps"]
    # the summary: R-hat of every dimension over the chains of ALL ranks (two all_gathers of 3*D+1 doubles,
    # summed in rank order), ESS and lane totals (one all_reduce each) -- timed on its own
    # (evaluated twice -- both are read-only: the first call of a process pays for loading the diagnostics' code
    # objects, ~0.1-0.2 s; the second is what a summary costs)
    def summary():
        rh_ = mom.rhat()
        e_ = rec.ess()
        e_ = torch.where(e_ > 0, e_, torch.full_like(e_, float(draws))).clamp(max=float(draws)).min(dim=0).values
        tot_ = bk.dist.sum_over_ranks(float(e_.sum().item()), ctx.device)
        lane_ = bk.dist.sum_over_ranks(float(lane_steps), ctx.device)
        torch.cuda.synchronize()
        return rh_, tot_, lane_

    ctx.barrier()
    t0 = time.perf_counter()
    summary()
    first_s = time.perf_counter() - t0
    calls0 = dict(bk.dist.collective_calls)
    ctx.barrier()
    t0 = time.perf_counter()
    rh, ess_total, lane_total = summary()
    summary_s = time.perf_counter() - t0
    calls = {k: bk.dist.collective_calls[k] - calls0[k] for k in calls0}
    flop_per_eval = 13.0 * D  # see DESIGN.md section 3: funnel gradient + kick + drift, per chain-step
    out = {"workload": "BASELINE.json configs[3]: Neal's funnel D=101, DRGHMC K=3 eps=(0.2,0.05,0.0125) L=(10,40,160) "
                       f"damping 0.1, {C} chains per GPU (global chain ids rank*{C}..), Welford "
                       f"R-hat over all dims and "
                       "ALL ranks' chains + ESS of 3 dims and logp",
           "bound": "fp64 VALU + exp latency (state register-resident inside a proposal; 26 MB "
                    "arrays are cache-resident)",
           "chains_per_gpu": C, "chains_total": C * ctx.world,
           "ms_per_draw": 1e3 * el / draws, "draws_per_sec": C * ctx.world * draws / el,
           "grad_evals_per_sec": lane_total / el, "mean_grad_evals_per_draw": lane_total / (C * ctx.world * draws),
           "fp64_tflops": lane_total * flop_per_eval / el / 1e12, "flop_model": "13*D flop per gradient evaluation",
           "rhat_max": float(rh.max()), "rhat_v": float(rh[0]), "ess_per_sec": ess_total / el, "draws": draws,
           "rhat_over_chains": C * ctx.world, "collectives_per_summary": calls, "summary_ms": 1e3 * summary_s,
           "summary_first_call_ms": 1e3 * first_s,
           "collective_backend": ctx.backend, "collective_ranks": ctx.collective_ranks(),
           "host_syncs_per_draw": getattr(s, "host_syncs_per_draw", None),
           "hipgraph": bool(getattr(s, "_use_graph", False)),
           "diagnostics": "Welford moments + tracked series updated inside the draw's hipGraph "
                          "(attach), no returned copies "
                          "(advance)", "timed_draws_follow_warmup_draws": warmup}
    if full_rhat:
        out["rhat"] = [float(v) for v in rh]
    # (extras: the provider variants of the same draws -- plugin, densities compiled from source, plain HMC on the funnel;
    # `bench.py --only cfg4` and `--full-secondary` run them, the default line keeps what SURVEY 8's rows need)
    if extras:
        try:
            out["model_opaque"] = bench_cfg4_model_opaque(ctx, C, D, warmup)
        except Exception as e:  # context only
            out["model_opaque"] = {"error": repr(e)}
        try:
            out["compiled_source_fused"] = bench_cfg4_compiled_source(ctx, C, D, warmup, draws)
        except Exception as e:  # context only
            out["compiled_source_fused"] = {"error": repr(e)}
        try:
            out["hmc_on_funnel"] = bench_hmc_lanes(ctx, C, D)
        except Exception as e:  # context only
            out["hmc_on_funnel"] = {"error": repr(e)}
    else:
        out["extras"] = "model_opaque / compiled_source_fused / hmc_on_funnel: python bench.py --only cfg4 (or --full-secondary)"
    if spec_length and ctx.world == 1 and C >= 32768:
        try:
            out["spec_length"] = bench_cfg4_spec_length(ctx, C, D)
            out["spec_length"]["note"] = ("v = theta[0] mixes slowly at these settings (mean ESS per "
                                          "chain above): from N(0, I) "
                                          "starts 1,100 draws do not reach v ~ N(0, 9); the run below "
                                          "starts from exact funnel draws")
            out["spec_length_stationary_start"] = bench_cfg4_spec_length(ctx, C, D, warmup=0, stationary_start=True)
        except Exception as e:  # context only
            out["spec_length"] = {"error": repr(e)}
    try:
        # The same sampler further into its run: the timed draws above are draws 4..43 from N(0, I) starts, where the
        # chains are still finding the funnel and the delayed-rejection stages run over 2-3x the lanes of the
        # stationary regime (mean_grad_evals_per_draw above against the one below).
        s.detach()
        mom2 = bk.RunningMoments(D, C)
        s.attach(moments=mom2)
        for _ in range(60):
            s.advance()
        b0 = float(s.lane_steps_total.item()) if on_device else 0.0
        chunks, per = 5, 20
        # (five timed chunks, the median reported: one replay in a few hundred stalls for tens of milliseconds inside
        # the HIP runtime -- seen as one 20-draw chunk at 2.07 instead of 0.34 ms per draw -- and would be 30 % of a
        # single 100-draw figure)
        els = sorted(ctx.timed_loop(s.advance, per) for _ in range(chunks))
        n2 = chunks * per
        ls2 = bk.dist.sum_over_ranks((float(s.lane_steps_total.item()) - b0) if on_device else float("nan"), ctx.device)
        out["after_100_draws"] = {"ms_per_draw": 1e3 * els[chunks // 2] / per,
            "ms_per_draw_slowest_chunk": 1e3 * els[-1] / per,
                                  "draws": n2, "chunks": chunks, "mean_grad_evals_per_draw": ls2 / (C * ctx.world * n2),
                                  "grad_evals_per_sec": ls2 / n2 * per / els[chunks // 2],
                                  "diagnostics": "Welford moments inside the draw"}
    except Exception as e:  # context only
        out["after_100_draws"] = {"error": repr(e)}
    return out


CFG4_ARGS = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)


def bench_cfg4_compiled_source(ctx, C, D, warmup, draws):
    """Config 4 with the funnel given as SOURCE (CTarget.from_source(form="lanes")): the generated translation unit
    instantiates the library's one-launch delayed-rejection proposal kernel (csrc/bk_lanes.hpp, the template bk.Funnel
    itself is an instantiation of) with the user's density inlined.  Same warm-up and draws as the built-in figure;
    must end bit-identical to it.  Also the stationary regime (draws 100..), like `after_100_draws`."""
    import torch

    import bayes_kit_amd as bk

    kw = dict(chains=C, chain_id0=ctx.rank * C, seed=20242)
    t0 = time.perf_counter()
    model = bk.CTarget.from_source(FUNNEL_LANES_SRC, D, form="lanes", head=1)
    build_s = time.perf_counter() - t0
    s = bk.DrGhmcDiag(model, *CFG4_ARGS, **kw)
    ref = bk.DrGhmcDiag(bk.Funnel(D), *CFG4_ARGS, **kw)
    for _ in range(warmup):
        s.sample()
        ref.sample()
    base = float(s.lane_steps_total.item())
    el = ctx.timed_loop(s.advance, draws)
    ls = float(s.lane_steps_total.item()) - base
    for _ in range(draws):
        ref.advance()
    same = bool(torch.equal(s._theta_dc, ref._theta_dc) and torch.equal(s._rho_dc, ref._rho_dc)
                and torch.equal(s._rng_state, ref._rng_state))
    out = {"what": "CTarget.from_source(<funnel as bk_lanes_density>, form='lanes', head=1): every "
                   "delayed-rejection proposal ONE "
                   "launch of the library's trajectory template with the compiled density inlined",
           "ms_per_draw": 1e3 * el / draws, "draws": draws, "grad_evals_per_sec": ls / el,
           "mean_grad_evals_per_draw": ls / (C * draws), "host_syncs_per_draw": s.host_syncs_per_draw,
           "hipgraph": bool(s._use_graph), "one_launch_proposals": bool(s._one_launch), "identical_to_builtin": same,
           "construction_s_incl_hipcc_or_cache": build_s}
    del ref
    for _ in range(60):
        s.advance()
    chunks, per = 5, 20
    els = sorted(ctx.timed_loop(s.advance, per) for _ in range(chunks))
    out["after_100_draws"] = {"ms_per_draw": 1e3 * els[chunks // 2] / per,
        "ms_per_draw_slowest_chunk": 1e3 * els[-1] / per,
                              "draws": chunks * per}
    return out


def bench_hmc_lanes(ctx, C, D, eps=0.05, L=32, draws=20):
    """Plain HMC (hmc.py:40-63) on the config-4 target through the lane-spread kernel templates: the whole trajectory
    as ONE launch (built-in and from source), ONE launch per leapfrog step, and the gradient as a separate op per step
    -- same draws."""
    import torch

    import bayes_kit_amd as bk

    kw = dict(chains=C, chain_id0=ctx.rank * C, seed=20244)
    res = {"workload": f"HMC eps={eps} L={L} on Neal's funnel D={D}, {C} chains per GPU",
           "bound": "fp64 VALU / launch latency"}
    ref = None
    for key, mk, k2 in (("one_launch_trajectory", lambda: bk.Funnel(D), {}),
                        ("one_launch_trajectory_from_source",
                         lambda: bk.CTarget.from_source(FUNNEL_LANES_SRC, D, form="lanes", head=1), {}),
                        ("one_launch_per_step", lambda: bk.Funnel(D), dict(path="step")),
                        ("gradient_separate_op", lambda: bk.Funnel(D), dict(path="opaque"))):
        s = bk.HMCDiag(mk(), eps, L, **kw, **k2)
        for _ in range(3):
            s.sample()
        el = ctx.timed_loop(s.sample, draws)
        res[key] = {"ms_per_draw": 1e3 * el / draws, "steps_per_sec": C * ctx.world * L * draws / el,
            "accept_rate": s.accept_rate(),
                    "hipgraph": bool(s._use_graph)}
        if ref is None:
            ref = s
        else:
            res[key]["identical_to_one_launch_trajectory"] = bool(torch.equal(s._theta_dc, ref._theta_dc)
                                                                  and torch.equal(s._rng_state, ref._rng_state))
            del s
    return res


def bench_cfg4_spec_length(ctx, C, D, draws=1000, warmup=100, stationary_start=False):
    """Config 4 AS SPECIFIED: N = 1,000 draws per chain after a burn-in, Welford R-hat over all 101
    dimensions (bayes_kit/rhat.py:163-171), ESS of dims {0, 1, 100} + the joint log density with its between-chain
    standard error (ess.py:52-69), and the funnel's marginal of v = theta_0 ~ N(0, 9) as a check of WHAT is sampled.
    Everything is fed from inside the draw's hipGraph; no state is copied out."""
    import torch

    import bayes_kit_amd as bk

    init = None
    if stationary_start:  # exact draws of the funnel: what 1,000 draws must leave invariant (tests/test_gpu_config4.py)
        g = torch.Generator().manual_seed(5)
        v0 = 3.0 * torch.randn(C, generator=g, dtype=torch.float64)
        rows = torch.exp(0.5 * v0)[:, None] * torch.randn((C, D - 1), generator=g, dtype=torch.float64)
        init = torch.cat([v0[:, None], rows], dim=1)
    s = bk.DrGhmcDiag(bk.Funnel(D), *CFG4_ARGS, chains=C, chain_id0=ctx.rank * C, seed=20242, init=init)
    for _ in range(warmup):
        s.advance()
    # The timed draws replay as graphs of `per` consecutive draws (DrGhmcDiag.advance(n): one graph launch per `per` draws).
    # attach() drops captured graphs, so `pre` untimed draws follow it: the eager / single-draw-graph ones a run starts
    # with and one replay of the multi-draw graph; they are recorded like the rest (the diagnostics below cover draws + pre).
    per = int(s.DRAWS_PER_GRAPH) if draws % int(s.DRAWS_PER_GRAPH) == 0 else 1
    pre = 2 + per
    mom = bk.RunningMoments(D, C)
    rec = bk.DrawRecorder([0, 1, D - 1], draws + pre, C)
    s.attach(moments=mom, recorder=rec)
    s.advance(pre)
    torch.cuda.synchronize()
    base = float(s.lane_steps_total.item())
    el = ctx.timed_loop(lambda: s.advance(per), draws // per)
    lane = float(s.lane_steps_total.item()) - base
    recorded = draws + pre
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rh = torch.as_tensor(mom.rhat())
    ess = rec.ess()                                    # [4 series, C]
    torch.cuda.synchronize()
    summary_s = time.perf_counter() - t0
    ess = torch.where(ess > 0, ess, torch.full_like(ess, float(recorded))).clamp(max=float(recorded))
    per_series = ess.sum(dim=1)
    mcse = (ess.var(dim=1, unbiased=True) * C).sqrt()  # between-chain standard error of each summed ESS
    ess_min = ess.min(dim=0).values
    v = rec.series[0, :recorded]                       # theta_0 of every chain, every draw  [draws, C]
    v_mean, v_var = float(v.mean()), float(v.var())
    ess_v = float(per_series[0])
    names = ["theta[0] (v)", "theta[1]", f"theta[{D - 1}]", "joint logp"]
    return {"what": f"configs[3] at its stated length: {draws} draws per chain after {warmup} "
                    f"burn-in draws, {C} chains, "
                    "diagnostics inside the draw's hipGraph; start: "
                    + ("exact draws of the funnel (invariance check)" if stationary_start
                       else "N(0, I) as the reference (hmc.py:24-28)"),
            "draws": draws, "burn_in": warmup, "untimed_recorded_draws": pre, "draws_per_graph_launch": per,
            "seconds": el, "ms_per_draw": 1e3 * el / draws,
            "grad_evals_per_sec": lane / el, "mean_grad_evals_per_draw": lane / (C * draws),
            "rhat_max": float(rh.max()), "rhat_v": float(rh[0]), "rhat_over_dims": int(rh.numel()),
            "ess_per_sec": {n: float(e) / el for n, e in zip(names, per_series)},
            "ess_per_sec_mcse": {n: float(m) / el for n, m in zip(names, mcse)},
            "ess_min_per_sec": float(ess_min.sum()) / el,
            "mean_ess_per_chain": {n: float(e) / C for n, e in zip(names, per_series)},
            "v_mean": v_mean, "v_mean_mcse": (9.0 / max(ess_v, 1.0)) ** 0.5, "v_var": v_var, "v_var_target": 9.0,
            "summary_ms": 1e3 * summary_s}


def bench_cfg4_model_opaque(ctx, C, D, warmup, draws=20):
    """Config 4 through the interface north_star names: the gradient a SEPARATE device op called once per leapfrog
    step (drghmc.py:280-283) -- the library's own funnel op with path="step", and a user plugin behind the
    counted plugin ABI (bk_target_fn_n) -- every lane count on the device, the draw one hipGraph.  Same warm-up as the
    fused figure above (draws 4..), so the two are comparable; the fused sampler run beside it must end
    bit-identical."""
    import torch

    import bayes_kit_amd as bk

    args = CFG4_ARGS
    kw = dict(chains=C, chain_id0=ctx.rank * C, seed=20242)
    plugin_lib = os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so")

    def run(model, n, **k2):
        so = bk.DrGhmcDiag(model, *args, **kw, **k2)
        for _ in range(warmup):
            so.sample()
        base = float(so.lane_steps_total.item()) if so._dev_counts else None
        el = ctx.timed_loop(so.advance, n)
        ls = (float(so.lane_steps_total.item()) - base) if so._dev_counts else float("nan")
        return so, {"ms_per_draw": 1e3 * el / n, "draws": n, "grad_evals_per_sec": ls / el,
                    "mean_grad_evals_per_draw": ls / (C * n), "host_syncs_per_draw": so.host_syncs_per_draw,
                    "hipgraph": bool(so._use_graph), "device_counts": bool(so._dev_counts)}

    ref, fused = run(bk.Funnel(D), draws)
    out = {"workload": "configs[3] with the gradient as a separate op per leapfrog step (one "
                       "counted gradient launch + one "
                       "counted kick+drift launch per step, 2*sum(L)+O(1) = ~580 launches per draw, "
                       "lane counts on the device)",
           "bound": "launch latency (dependent chain of ~580 small launches per draw inside one hipGraph)",
           "fused_one_launch_proposals_same_draws": fused}
    so, r = run(bk.Funnel(D), draws, path="opaque")
    r["identical_to_fused"] = bool(torch.equal(so._theta_dc, ref._theta_dc) and torch.equal(so._rho_dc, ref._rho_dc)
                                   and torch.equal(so._rng_state, ref._rng_state))
    out["builtin_gradient_op"] = r
    del so
    # the same path with a leapfrog step {gradient, kick, drift} as ONE launch (bk_leapfrog_step_funnel: the model's
    # density inside the library's step kernel, csrc/bk_lanes.hpp) -- what the step-by-step path runs by default for a
    # model that has it
    so, r = run(bk.Funnel(D), draws, path="step")
    r["identical_to_fused"] = bool(torch.equal(so._theta_dc, ref._theta_dc) and torch.equal(so._rho_dc, ref._rho_dc)
                                   and torch.equal(so._rng_state, ref._rng_state))
    out["builtin_one_launch_steps"] = r
    del so
    if os.path.exists(plugin_lib):
        so, r = run(bk.CTarget(plugin_lib, "funnel_target", D, counted_symbol="funnel_target_n"), draws)
        r["identical_to_fused"] = bool(torch.equal(so._theta_dc, ref._theta_dc) and torch.equal(so._rho_dc, ref._rho_dc)
                                       and torch.equal(so._rng_state, ref._rng_state))
        out["plugin_ctarget"] = r
        del so
    # the same density from SOURCE on the counted path: form="lanes" (a chain spread over 4 / 8 / 16 lanes, DPP sums;
    # path="step" keeps it off its one-launch path) and form="chain" (one lane per chain walking D coordinates)
    # ("_one_launch_steps": the lanes form also gives the step-by-step path {gradient, kick, drift} as ONE launch per
    # leapfrog step)
    lanes = dict(form="lanes", head=1)
    # ("_one_launch_trajectories": the per-chain form -- ANY coupled density -- runs the whole trajectory of a proposal as one
    # launch, theta in its lane's registers and rho in LDS through all the steps, followed by the library's finish launch)
    off = dict(path="step")
    for key, src, kws, k2 in (("compiled_source_lanes", FUNNEL_LANES_SRC, lanes, dict(off, path="opaque")),
                              ("compiled_source_lanes_one_launch_steps", FUNNEL_LANES_SRC, lanes, off),
                              ("compiled_source_chain", FUNNEL_CHAIN_SRC, dict(form="chain"), dict(off, path="opaque")),
                              ("compiled_source_chain_one_launch_steps", FUNNEL_CHAIN_SRC, dict(form="chain"), off),
                              ("compiled_source_chain_one_launch_trajectories", FUNNEL_CHAIN_SRC, dict(form="chain"), {})):
        try:
            so, r = run(bk.CTarget.from_source(src, D, **kws), draws, **k2)
            same = torch.equal(so._theta_dc, ref._theta_dc) and torch.equal(so._rho_dc, ref._rho_dc) and \
                torch.equal(so._rng_state, ref._rng_state)
            # (the one-lane-per-chain form sums the coordinates in order: a different, equally valid rounding of s)
            r["identical_to_fused"] = bool(same)
            out[key] = r
            del so
        except Exception as e:  # context only
            out[key] = {"error": repr(e)}
    # the same draws with launches sized by host reads (three lane counts read back per draw; the path every model
    # had before the counted entry points, and the one PyTorch-autograd models still take)
    so, r = run(bk.Funnel(D), 5, path="opaque", device_counts=False)
    out["host_sized_launches"] = r
    out["ms_per_draw"] = out["builtin_gradient_op"]["ms_per_draw"]
    return out


def bench_mala(ctx, draws=20, warmup=3, chains=C_CFG3):
    """MALA at config-3 shape: 88*D algorithmic bytes per chain-draw with the model's gradient a separate op; beside
    it the same draws with the separable density inlined into the step kernel (56*D), reported on its own model."""
    import torch

    import bayes_kit_amd as bk

    C, D = chains, D_CFG3
    lam = torch.logspace(0, 4, D, dtype=torch.float64)

    def run(**kw):
        s = bk.MALA(bk.DiagGaussian(lam), 5e-5, chains=C, chain_id0=ctx.rank * C, seed=7, **kw)
        s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(ctx.device)[:, None])
        s.refresh_cache()
        for _ in range(warmup):
            s.sample()
        return s, ctx.timed_loop(s.sample, draws) / draws

    s, per = run(path="step")
    out = {"workload": "MALA eps=5e-5 on the config-3 target (D=1024, 65,536 chains per GPU), "
                       "model-opaque gradient op",
           "bound": "hbm", "ms_per_draw": 1e3 * per, "draws_per_sec": C * ctx.world / per,
           "algorithmic_bytes_per_chain_draw": 88 * D, "achieved": 88.0 * D * C / per / 1e9, "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": 88.0 * D * C / per / 1e9 / HBM_PEAK_GBPS, "accept_rate": s.accept_rate(),
           "path": getattr(s, "path", None), "placement": s.placement}
    last = s._theta_dc.clone()
    del s
    f, per_f = run()
    out["density_inlined"] = {
        "what": "the same draws with the separable density inlined into the step kernel (model.bk_mala_step: both gradients "
                "recomputed, none stored; the model's launch is its log density alone)",
        "ms_per_draw": 1e3 * per_f, "draws_per_sec": C * ctx.world / per_f, "algorithmic_bytes_per_chain_draw": 56 * D,
        "achieved": 56.0 * D * C / per_f / 1e9, "unit": "GB/s", "frac_56D_model": 56.0 * D * C / per_f / 1e9 / HBM_PEAK_GBPS,
        "accept_rate": f.accept_rate(), "path": f.path, "identical_to_model_opaque": bool(torch.equal(f._theta_dc, last))}
    return out


def bench_torch_model(ctx, draws=2, warmup=1, chains=C_CFG3, extras=True):
    """Config-3 workload with the gradient supplied by user PyTorch code through autograd."""
    import torch

    import bayes_kit_amd as bk

    C, D, L = chains, D_CFG3, L_CFG3
    lam = torch.logspace(0, 4, D, dtype=torch.float64, device=ctx.device)
    model = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), D)
    s = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                   metric_diag=torch.ones(D, dtype=torch.float64), tune_placement=False)
    s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
    for _ in range(warmup):
        s.sample()
    el = ctx.timed_loop(s.sample, draws)
    per = el / draws
    out = {"workload": "config-3 shape, gradient = torch autograd of a user log density (TorchModel)",
           "bound": "hbm (the model's own temporaries and extra passes, not the integrator)",
           "ms_per_draw": 1e3 * per, "steps_per_sec": C * ctx.world * L / per,
           "path_hbm_frac_56D_model": C * L / per * 56.0 * D / 1e9 / HBM_PEAK_GBPS, "accept_rate": s.accept_rate()}
    # the same density written for the engine's own (D, C) layout: torch's contiguous kernels, a chain-contiguous
    # gradient
    del s
    model = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), D, layout="dc")
    s = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                   metric_diag=torch.ones(D, dtype=torch.float64), tune_placement=False)
    s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
    for _ in range(warmup):
        s.sample()
    per_dc = ctx.timed_loop(s.sample, draws) / draws
    out["engine_layout"] = {"what": "TorchModel(fn, D, layout='dc'): fn takes the (D, C) array",
        "ms_per_draw": 1e3 * per_dc,
                            "steps_per_sec": C * ctx.world * L / per_dc,
                            "path_hbm_frac_56D_model": C * L / per_dc * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
                            "accept_rate": s.accept_rate()}
    del s
    # ... and with the gradient written out in torch ops as well (TorchModel(grad_fn=...)): no autograd graph, a
    # leapfrog step is one torch kernel beside the engine's kick+drift
    try:
        model = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), D, layout="dc",
                              grad_fn=lambda Th: -(lam[:, None] * Th))
        s = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                       metric_diag=torch.ones(D, dtype=torch.float64), tune_placement=False)
        s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
        for _ in range(warmup):
            s.sample()
        per_g = ctx.timed_loop(s.sample, draws) / draws
        out["written_out_gradient"] = {"what": "TorchModel(fn, D, layout='dc', grad_fn=...): the gradient "
                                               "as torch ops, no autograd",
                                       "ms_per_draw": 1e3 * per_g, "steps_per_sec": C * ctx.world * L / per_g,
                                       "path_hbm_frac_56D_model": C * L / per_g * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
                                       "accept_rate": s.accept_rate()}
        del s
    except Exception as e:  # context only
        out["written_out_gradient"] = {"error": repr(e)}
    # the same density as six lines of HIP C++ handed to CTarget.from_source: compiled with hipcc at construction into
    # the plugin ABI (bk_target_fn / bk_target_fn_n) -- what a Python user reaches without writing a build
    try:
        src = ("__device__ __forceinline__ void bk_term(double th, i64 d, const double* lam, "
               "double& term, double& grad) {\n"
               "  const double t = lam[d] * th;\n  term = -0.5 * (th * t);\n  grad = -t;\n}\n")
        t0 = time.perf_counter()
        model = bk.CTarget.from_source(src, D, params=lam)
        build_s = time.perf_counter() - t0
        s = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                       # (model-opaque: the gradient a separate op per step, priced on the 56 D model like the headline)
                       metric_diag=torch.ones(D, dtype=torch.float64), path="opaque")
        s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
        for _ in range(warmup + 1):
            s.sample()
        n = max(draws, 5)
        per_src = ctx.timed_loop(s.sample, n) / n
        out["compiled_source"] = {"what": "CTarget.from_source(<6 lines of HIP C++>, "
                                          "form='elementwise'): hipcc at construction, "
                                          "plugin ABI, streaming 16-byte-per-lane gradient kernel",
                                  "ms_per_draw": 1e3 * per_src, "steps_per_sec": C * ctx.world * L / per_src,
                                  "path_hbm_frac_56D_model": C * L / per_src * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
                                  "accept_rate": s.accept_rate(), "construction_s_incl_hipcc_or_cache": build_s}
    except Exception as e:  # context only
        out["compiled_source"] = {"error": repr(e)}
    # TorchModel(compile=True): the SAME PyTorch lambda as the autograd figure above, read once with torch.fx, its
    # per-coordinate term and hand-differentiated derivative emitted as bk_term source and compiled (trace.py):
    # model-opaque step-by-step path (path="step": priced on the 56*D model like the headline) and the whole-draw
    # kernel it also unlocks (separately)
    try:
        t0 = time.perf_counter()
        model = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), D, compile=True)
        build_s = time.perf_counter() - t0
        if model.compiled is None:
            raise RuntimeError("not traced: " + str(model.compile_note))
        s = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                       metric_diag=torch.ones(D, dtype=torch.float64), path="opaque")
        s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
        for _ in range(warmup + 1):
            s.sample()
        n = max(draws, 5)
        per_t = ctx.timed_loop(s.sample, n) / n
        # gradient against autograd on fresh points
        Th = torch.randn((4096, D), dtype=torch.float64, device=ctx.device) / torch.sqrt(lam)
        x = Th.clone().requires_grad_(True)
        (g_auto,) = torch.autograd.grad((-0.5 * (x * x * lam).sum(dim=1)).sum(), x)
        _, g_c = model.log_density_gradient(Th)
        rel = float(((g_c - g_auto).abs().max() / g_auto.abs().max()).item())
        out["traced_source"] = {"what": "TorchModel(fn, D, compile=True): the PyTorch lambda traced with "
                                        "torch.fx, bk_term (value + "
                                        "derivative) generated and compiled; step-by-step model-opaque path",
                                "ms_per_draw": 1e3 * per_t, "steps_per_sec": C * ctx.world * L / per_t,
                                "path_hbm_frac_56D_model": C * L / per_t * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
                                "accept_rate": s.accept_rate(), "gradient_max_rel_err_vs_autograd": rel,
                                "construction_s_incl_trace_hipcc_or_cache": build_s}
        del s
        # (the same model with {gradient, kick, drift} as ONE launch per leapfrog step: 32 D bytes per chain-step
        # instead of 56 D, so NOT on the 56 D model either)
        h1 = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                        metric_diag=torch.ones(D, dtype=torch.float64), path="step")
        h1._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
        for _ in range(warmup + 1):
            h1.sample()
        per_h = ctx.timed_loop(h1.sample, n) / n
        out["traced_source"]["one_launch_per_step"] = {
            "what": "bk_leapfrog_step: gradient + kick + drift in one streaming launch (32 D "
                    "algorithmic bytes per chain-step)",
            "ms_per_draw": 1e3 * per_h, "steps_per_sec": C * ctx.world * L / per_h, "step_hook": bool(h1._step_hook),
            "hbm_frac_32D_model": C * L / per_h * 32.0 * D / 1e9 / HBM_PEAK_GBPS}
        del h1
        f = bk.HMCDiag(model, EPS_CFG3, L, chains=C, chain_id0=ctx.rank * C, seed=SEED_CFG3,
                       metric_diag=torch.ones(D, dtype=torch.float64))
        f._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
        for _ in range(3):
            f.sample()
        per_f = ctx.timed_loop(f.sample, 20) / 20
        out["traced_source"]["whole_draw_kernel"] = {"what": "the same model through the whole-draw kernel "
                                                             "(fp64-VALU bound; NOT on "
                                                             "the 56*D model)", "ms_per_draw": 1e3 * per_f,
                                                     "steps_per_sec": C * ctx.world * L / per_f,
                                                     "fused_draw": bool(f._fused_draw)}
        del f
    except Exception as e:  # context only
        out["traced_source"] = {"error": repr(e)}
    # ... and a HIERARCHICAL density written in PyTorch (Neal's funnel, config-4 shape): traced into the lane-spread
    # form (trace_lanes.py), every delayed-rejection proposal one launch -- against the same function through autograd
    if not extras:
        out["extras"] = "traced_hierarchical / traced_coupled (the tracers beyond a separable density): python bench.py --only torch_model (or --full-secondary)"
        return out
    try:
        Df, Cf = 101, 32768

        def funnel(Th):
            v, x = Th[:, 0], Th[:, 1:]
            return -(v * v) / 18.0 - 0.5 * (Df - 1) * v - 0.5 * torch.exp(-v) * (x * x).sum(dim=1)

        args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)
        t0 = time.perf_counter()
        traced = bk.TorchModel(funnel, Df, compile=True)
        build_s = time.perf_counter() - t0
        s = bk.DrGhmcDiag(traced, *args, chains=Cf, chain_id0=ctx.rank * Cf, seed=20242)
        for _ in range(3):
            s.advance()
        per_l = ctx.timed_loop(s.advance, 40) / 40
        a = bk.DrGhmcDiag(bk.TorchModel(funnel, Df), *args, chains=Cf, chain_id0=ctx.rank * Cf, seed=20242)
        for _ in range(2):
            a.sample()
        per_a = ctx.timed_loop(a.sample, 3) / 3
        out["traced_hierarchical"] = {"what": "Neal's funnel D=101 as a PyTorch function (Th[:, 0], Th[:, 1:], "
                                              "a sum over rows), DRGHMC "
                                              "K=3 on 32,768 chains: TorchModel(compile=True) -> lanes form "
                                              "-> one launch per proposal",
                                      "compiled_form": getattr(traced, "compiled_form", None),
                                      "ms_per_draw": 1e3 * per_l,
                                      "one_launch_proposals": bool(s._one_launch),
                                      "host_syncs_per_draw": s.host_syncs_per_draw,
                                      "autograd_ms_per_draw": 1e3 * per_a,
                                      "autograd_host_syncs_per_draw": a.host_syncs_per_draw,
                                      "speedup_vs_autograd": per_a / per_l,
                                      "construction_s_incl_trace_hipcc_or_cache": build_s}
        del s, a
    except Exception as e:  # context only
        out["traced_hierarchical"] = {"error": repr(e)}
    # (3) a density whose coordinates are COUPLED through shifted slices: an AR(1) state-space model (learned correlation and
    # innovation scale, Gaussian observations) written with slices in PyTorch -> trace_chain.py -> the per-chain form: the whole
    # trajectory of a proposal one launch -- against the same function through autograd
    try:
        Da, Ca = 101, 32768  # (config 4's shape)
        ya = torch.sin(torch.linspace(0.0, 9.0, Da - 2, dtype=torch.float64, device=ctx.device))

        def ar1(Th):
            phi, ls, x = torch.tanh(Th[:, 0]), Th[:, 1], Th[:, 2:]
            inn = x[:, 1:] - phi[:, None] * x[:, :-1]
            return -0.5 * (inn * inn).sum(-1) * torch.exp(-2 * ls) - (Da - 3) * ls \
                - 0.5 * x[:, 0] ** 2 * (1 - phi * phi) * torch.exp(-2 * ls) - 2.0 * ((ya - x) ** 2).sum(-1) \
                - 0.5 * Th[:, 0] ** 2 - 0.5 * (ls + 1.0) ** 2 / 0.09

        args = (3, [0.03, 0.012, 0.005], [10, 40, 160], 0.1)
        th0 = 0.3 * torch.randn((Ca, Da), dtype=torch.float64, device=ctx.device)
        t0 = time.perf_counter()
        tm = bk.TorchModel(ar1, Da, compile=True)
        build_s = time.perf_counter() - t0
        rec = {"what": "AR(1) state-space model D=101 written with shifted slices in PyTorch, DRGHMC K=3 L=(10,40,160) on 32,768 "
                       "chains: TorchModel(compile=True) -> per-chain form -> one launch per trajectory",
               "compiled_form": getattr(tm, "compiled_form", None), "construction_s_incl_trace_hipcc_or_cache": build_s}
        for key, kw in (("ms_per_draw", {}), ("ms_per_draw_one_launch_per_step", dict(path="step")),
                        ("ms_per_draw_gradient_separate_op", dict(path="opaque"))):
            s = bk.DrGhmcDiag(bk.TorchModel(ar1, Da, compile=True), *args, chains=Ca, seed=20246, init=th0, **kw)
            for _ in range(3):
                s.sample()
            rec[key] = 1e3 * ctx.timed_loop(s.advance, 10) / 10
            rec["host_syncs_per_draw"] = s.host_syncs_per_draw
            del s
        a = bk.DrGhmcDiag(bk.TorchModel(ar1, Da), *args, chains=Ca, seed=20246, init=th0)
        a.sample()
        per_a = ctx.timed_loop(a.sample, 2) / 2
        rec.update(autograd_ms_per_draw=1e3 * per_a, autograd_host_syncs_per_draw=a.host_syncs_per_draw,
                   speedup_vs_autograd=1e3 * per_a / rec["ms_per_draw"])
        out["traced_coupled"] = rec
        del a
    except Exception as e:  # context only
        out["traced_coupled"] = {"error": repr(e)}
    return out


def bench_cfg5(ctx, N=1_000_000, D=512, chains=2048, grad_reps=3):
    """configs[4]: synthetic logistic regression N=1e6, D=512, 2,048 chains: the gradient for all chains
    (two fp64 MFMA GEMMs + one elementwise pass), one HMC draw with a dense metric, one temperature of the
    likelihood-annealed SMC (smc.py:47-75).  No reference oracle (parity by tolerance in tests/); priced
    against the dense fp64 MFMA peak."""
    import torch

    import bayes_kit_amd as bk

    dev, C = ctx.device, chains
    g = torch.Generator(device=dev)
    g.manual_seed(20243)
    X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=g) / D ** 0.5
    tstar = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
    y = (torch.rand(N, dtype=torch.float64, device=dev, generator=g) < torch.sigmoid(X @ tstar)).to(torch.float64)
    model = bk.LogisticRegression(X, y, prior_scale=1.0)
    th = torch.randn((D, C), dtype=torch.float64, device=dev, generator=g) * 0.1
    grad, lp = torch.empty_like(th), torch.empty(C, dtype=torch.float64, device=dev)
    model.bk_eval(th, grad, lp)  # warm-up: uploads X^T, sizes the scratch
    el = ctx.timed_loop(lambda: model.bk_eval(th, grad, lp), grad_reps) / grad_reps
    el_g = ctx.timed_loop(lambda: model.bk_eval(th, grad, None), grad_reps) / grad_reps  # what a leapfrog step asks for
    flop = 2 * 2.0 * N * D * C  # Z = X Theta and G = X^T R
    out = {"workload": f"BASELINE.json configs[4]: logistic regression N={N} D={D}, {C} chains "
                       f"per GPU (synthetic, torch "
                       "seed 20243), gradient = 2 fp64 MFMA GEMMs + residual pass",
           "bound": "mfma", "gradient_ms": 1e3 * el, "gradient_evals_per_sec": C * ctx.world / el,
           "achieved": flop / el / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": flop / el / 1e12 / FP64_MFMA_PEAK_TFLOPS, "flop_model": "4*N*D flop per chain-gradient",
           "gradient_only": {"what": "the same without the log density (a leapfrog step's call)", "ms": 1e3 * el_g,
                             "achieved": flop / el_g / 1e12, "frac": flop / el_g / 1e12 / FP64_MFMA_PEAK_TFLOPS}}
    # one draw of HMC with a dense metric (velocity covariance ~ the posterior scale 4 D / N)
    L = 4
    Md = torch.eye(D, dtype=torch.float64) * (4.0 * D / N)
    s = bk.HMCDiag(model, 0.3, L, chains=C, chain_id0=ctx.rank * C, seed=20243, metric_dense=Md,
                   init=th.t().contiguous().cpu())
    s.sample()
    el = ctx.timed_loop(s.sample, 1)
    # per step: the gradient (4 N D) + M @ grad (2 D^2); per draw also chol(M) z and M^-1 rho twice
    hflop = C * (L * (4.0 * N * D + 2.0 * D * D) + 3 * 2.0 * D * D)
    out["hmc_dense_metric"] = {"leapfrog_steps": L, "ms_per_draw": 1e3 * el, "steps_per_sec": C * ctx.world * L / el,
                               "tflops_fp64": hflop / el / 1e12,
                               "frac_of_mfma_peak": hflop / el / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                               "accept_rate": s.accept_rate()}
    del s
    # The reference's ladder t = n / N (smc.py:42-43) at this size, for the record: the first reweighting of an 8-step
    # ladder keeps one or two particles (round 3 timed a temperature of that collapsed system).
    init = torch.randn((C, D), dtype=torch.float64, device=dev, generator=g)
    smc = bk.TemperedLikelihoodSMC(model, C, 8, init, bk.hmc_kernel(0.5, 2, metric_dense=Md), seed=20243)
    smc.transition(1)
    out["fixed_ladder_n_over_8"] = {"ess_after_first_reweighting": float(smc.last_ess), "particles": C,
                                    "note": "t = n/N as the reference (smc.py:42-43): degenerate at 1e6 observations"}
    del smc
    # Config 5 END TO END: the whole annealed ladder, temperatures chosen so that every reweighting keeps half the
    # particles (adaptive = 0.5, an extension marked as such in smc.py), one HMC move (L = 2) per temperature under a
    # dense metric re-estimated from the particles, log-sum-exp weights, multinomial resampling (smc.py:45-75).
    ess_target, L_smc, eps_smc = 0.5, 2, 0.4
    kern = bk.hmc_kernel(eps_smc, L_smc, adapt_metric=True)
    smc = bk.TemperedLikelihoodSMC(model, C, 1, init, kern, seed=20243, adaptive=ess_target)
    ctx.barrier()
    t0 = time.perf_counter()
    smc.run()
    torch.cuda.synchronize()
    ctx.barrier()
    el = time.perf_counter() - t0
    T = smc.temperatures
    post = smc.thetas.mean(dim=0)
    # evaluations of all particles: L per move (+ one at the very first temperature; afterwards the log density and
    # gradient at a new temperature come from the untempered parts kept with each particle: bk_retemper, no pass over
    # the data)
    evals = len(T) * L_smc + 1
    out["annealed_smc"] = {"particles": C,
        "ladder": f"adaptive, ESS target {ess_target} x particles (extension; the reference "
                                                     "has t = n/N)", "temperatures": len(T),
                           "first_temperatures": T[:3], "min_ess_over_ladder": min(smc.ess_history),
                           "move": f"HMC eps={eps_smc} L={L_smc}, dense metric = particle variances, "
                                   f"re-estimated per temperature",
                           "accept_rate_min": min(kern.accept_rates),
                           "accept_rate_mean": sum(kern.accept_rates) / len(T),
                           "seconds": el, "model_evaluations": evals,
                           "tflops_fp64": evals * 4.0 * N * D * C / el / 1e12,
                           "frac_of_mfma_peak": evals * 4.0 * N * D * C / el / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                           "corr_posterior_mean_vs_truth":
                               float(torch.corrcoef(torch.stack([post, tstar]))[0, 1].item()),
                           "rel_err_posterior_mean_vs_truth": float(((post - tstar).norm() / tstar.norm()).item())}
    return out


def run_secondary(ctx, which, **kw):
    import torch

    table = {"cfg2": bench_cfg2, "cfg4": bench_cfg4, "mala": bench_mala, "torch_model": bench_torch_model,
             "cfg5": bench_cfg5}
    out = {}
    extras = kw.pop("extras", True)
    for name in which:
        t0 = time.perf_counter()
        try:
            out[name] = table[name](ctx, **dict(kw, extras=extras) if name in ("cfg4", "torch_model") else kw)
        except Exception as e:  # a secondary figure must never cost the headline line
            out[name] = {"error": repr(e)}
        out[name]["bench_wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()
    return out
