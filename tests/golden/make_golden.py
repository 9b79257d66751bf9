#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (flatironinstitute/bayes-kit).

Build-container only: imports ``bayes_kit`` from ``/root/reference`` (read-only; never
copied, never shipped) and runs its samplers / diagnostics on seeded inputs.  Outputs are
small ``.npz`` fixtures (inputs + expected outputs) committed under ``tests/golden/``.

Seeding: every chain c of a case is the reference sampler constructed with
``seed=np.random.Philox(key=[case_seed, c])`` (``bayes_kit/typing.py:11`` admits a
BitGenerator), which is exactly the per-chain device stream of the HIP engine.  Cases
with ``pcg_seed`` use an int seed (PCG64) like the reference's README and tests.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import bayes_kit as ref  # noqa: E402  (the reference)
from oracle import models  # noqa: E402  (targets are user code from the reference's view)


def make_model(spec):
    kind = spec["kind"]
    if kind == "std_normal":
        return models.StdNormal()
    if kind == "iso_gaussian":
        return models.IsoGaussian(spec["D"])
    if kind == "diag_gaussian":
        return models.DiagGaussian(np.logspace(spec["log10_lo"], spec["log10_hi"], spec["D"]))
    if kind == "funnel":
        return models.Funnel(spec["D"])
    if kind == "ref_binomial":
        # the reference's OWN test model (test/models/binomial.py), imported from /root/reference
        from test.models.binomial import Binomial

        return Binomial(alpha=2, beta=3, x=5, N=15)
    raise KeyError(kind)


def make_metric(spec, D):
    if spec is None:
        return None
    if spec["kind"] == "linspace":
        return np.linspace(spec["lo"], spec["hi"], D)
    if spec["kind"] == "ones":
        return np.ones(D)
    raise KeyError(spec)


class CountingModel:
    """Counts gradient evaluations per draw (control-flow fixture for DRGHMC)."""

    def __init__(self, inner):
        self._inner = inner
        self.grad_calls = 0

    def dims(self):
        return self._inner.dims()

    def log_density(self, theta):
        return self._inner.log_density(theta)

    def log_density_gradient(self, theta):
        self.grad_calls += 1
        return self._inner.log_density_gradient(theta)


def rng_state(gen):
    st = gen.bit_generator.state
    if st["bit_generator"] == "Philox":
        return np.concatenate(
            [
                np.asarray(st["state"]["key"], dtype=np.uint64),
                np.asarray(st["state"]["counter"], dtype=np.uint64),
                np.asarray(st["buffer"], dtype=np.uint64),
                np.asarray([st["buffer_pos"]], dtype=np.uint64),
            ]
        )
    s = st["state"]  # PCG64: 128-bit state and increment as (hi, lo) words
    return np.asarray(
        [s["state"] >> 64, s["state"] & (2**64 - 1), s["inc"] >> 64, s["inc"] & (2**64 - 1)],
        dtype=np.uint64,
    )


def make_proposal(spec, c):
    """User-side proposal callbacks (the reference takes them as arguments, metropolis.py:83-84):
    theta* ~ N(a * theta, scale^2 I) from the proposal's OWN per-chain Philox stream."""
    prng = np.random.Generator(np.random.Philox(key=[spec["seed"], c]))
    a, scale = spec.get("a", 1.0), spec["scale"]

    def proposal_fn(theta):
        return prng.normal(loc=a * theta, scale=scale)

    def transition_lp_fn(to, frm):
        r = (to - a * frm) / scale
        return -0.5 * np.sum(r * r)

    return proposal_fn, transition_lp_fn


def run_sampler_case(case):
    C, N = case["chains"], case["draws"]
    model0 = make_model(case["model"])
    D = model0.dims()
    draws = np.empty((N, C, D))
    logps = np.empty((N, C))
    grad_calls = np.zeros((N, C), dtype=np.int64)
    theta0 = np.empty((C, D))
    states = []
    rho_final = np.zeros((C, D))
    for c in range(C):
        model = CountingModel(make_model(case["model"]))
        if "pcg_seed" in case:
            seed = case["pcg_seed"] + c
        else:
            seed = np.random.Philox(key=[case["seed"], c])
        init = None
        if case.get("init") is not None:
            init = np.asarray(case["init"], dtype=np.float64).copy()
        alg = case["alg"]
        if alg == "hmc":
            s = ref.HMCDiag(model, stepsize=case["stepsize"], steps=case["steps"], init=init, seed=seed)
        elif alg == "mala":
            s = ref.MALA(model, epsilon=case["epsilon"], init=init, seed=seed)
        elif alg == "drghmc":
            s = ref.DrGhmcDiag(
                model,
                max_proposals=case["max_proposals"],
                leapfrog_step_sizes=case["leapfrog_step_sizes"],
                leapfrog_step_counts=case["leapfrog_step_counts"],
                damping=case["damping"],
                init=init,
                seed=seed,
                prob_retry=case.get("prob_retry", True),
            )
        elif alg in ("metropolis", "mh"):
            proposal_fn, transition_lp_fn = make_proposal(case["proposal"], c)
            if alg == "metropolis":
                s = ref.Metropolis(model, proposal_fn, init=init, seed=seed)
            else:
                s = ref.MetropolisHastings(model, proposal_fn, transition_lp_fn, init=init, seed=seed)
        else:
            raise KeyError(alg)
        metric = make_metric(case.get("metric"), D)
        if metric is not None:
            s._metric = metric  # the reference cannot take a D>1 metric in its constructor
        theta0[c] = np.asarray(s._theta, dtype=np.float64)
        for n in range(N):
            before = model.grad_calls
            th, lp = s.sample()
            draws[n, c] = th
            logps[n, c] = lp
            grad_calls[n, c] = model.grad_calls - before
        states.append(rng_state(s._rng))
        if alg == "drghmc":
            rho_final[c] = s._rho
    return dict(
        case=np.array(json.dumps(case)),
        theta0=theta0,
        draws=draws,
        logp=logps,
        grad_calls=grad_calls,
        rng_state=np.stack(states),
        rho_final=rho_final,
    )


SAMPLER_CASES = [
    # --- HMC (bayes_kit/hmc.py) ---
    dict(name="hmc_stdnormal", alg="hmc", model=dict(kind="std_normal"), stepsize=0.25, steps=10,
         chains=4, draws=50, seed=101),
    dict(name="hmc_steps0", alg="hmc", model=dict(kind="std_normal"), stepsize=0.25, steps=0,
         chains=2, draws=10, seed=102),
    dict(name="hmc_iso4", alg="hmc", model=dict(kind="iso_gaussian", D=4), stepsize=0.3, steps=5,
         chains=8, draws=50, seed=103),
    dict(name="hmc_iso128_cfg2", alg="hmc", model=dict(kind="iso_gaussian", D=128), stepsize=0.05,
         steps=32, chains=8, draws=20, seed=20240, metric=dict(kind="ones")),
    dict(name="hmc_diag16_metric", alg="hmc",
         model=dict(kind="diag_gaussian", D=16, log10_lo=0, log10_hi=1), stepsize=0.05, steps=8,
         chains=8, draws=40, seed=104, metric=dict(kind="linspace", lo=0.5, hi=1.5)),
    dict(name="hmc_diag1024_cfg3", alg="hmc",
         model=dict(kind="diag_gaussian", D=1024, log10_lo=0, log10_hi=4), stepsize=0.006, steps=64,
         chains=2, draws=4, seed=20241, metric=dict(kind="ones")),
    dict(name="hmc_pcg_seed", alg="hmc", model=dict(kind="iso_gaussian", D=3), stepsize=0.3, steps=4,
         chains=2, draws=25, pcg_seed=123),
    # --- MALA (bayes_kit/mala.py) ---
    dict(name="mala_readme_cfg1", alg="mala", model=dict(kind="std_normal"), epsilon=0.2,
         chains=1, draws=1000, pcg_seed=12345),
    dict(name="mala_stdnormal", alg="mala", model=dict(kind="std_normal"), epsilon=0.2,
         chains=4, draws=100, seed=201),
    dict(name="mala_iso8", alg="mala", model=dict(kind="iso_gaussian", D=8), epsilon=0.1,
         chains=8, draws=60, seed=202),
    dict(name="mala_diag16", alg="mala",
         model=dict(kind="diag_gaussian", D=16, log10_lo=0, log10_hi=1), epsilon=0.02,
         chains=8, draws=60, seed=203),
    dict(name="mala_diag48", alg="mala",
         model=dict(kind="diag_gaussian", D=48, log10_lo=0, log10_hi=1), epsilon=0.01,
         chains=4, draws=40, seed=205),  # D >= 32: the wavefront-per-chain generator's path
    dict(name="mala_init", alg="mala", model=dict(kind="iso_gaussian", D=3), epsilon=0.15,
         init=[0.2, -1.0, 0.5], chains=3, draws=30, seed=204),
    # --- the reference's own Binomial test model (scipy densities, finite-difference gradient) ---
    dict(name="hmc_ref_binomial", alg="hmc", model=dict(kind="ref_binomial"), stepsize=0.08, steps=3,
         init=[0.1], chains=2, draws=40, seed=401),
    dict(name="mala_ref_binomial", alg="mala", model=dict(kind="ref_binomial"), epsilon=0.12,
         init=[0.1], chains=2, draws=40, seed=402),
    dict(name="drghmc_ref_binomial", alg="drghmc", model=dict(kind="ref_binomial"), max_proposals=2,
         leapfrog_step_sizes=[0.3, 0.1], leapfrog_step_counts=[2, 6], damping=0.3, init=[0.1],
         chains=2, draws=40, seed=403),
    # --- DRGHMC (bayes_kit/drghmc.py) ---
    dict(name="drghmc_stdnormal_k3", alg="drghmc", model=dict(kind="std_normal"), max_proposals=3,
         leapfrog_step_sizes=[0.9, 0.45, 0.225], leapfrog_step_counts=[2, 4, 8], damping=0.2,
         chains=4, draws=200, seed=301),
    dict(name="drghmc_iso4_k2_noretry", alg="drghmc", model=dict(kind="iso_gaussian", D=4),
         max_proposals=2, leapfrog_step_sizes=[1.2, 0.4], leapfrog_step_counts=[2, 6], damping=1.0,
         prob_retry=False, chains=4, draws=100, seed=302),
    dict(name="drghmc_k1", alg="drghmc", model=dict(kind="iso_gaussian", D=5), max_proposals=1,
         leapfrog_step_sizes=[0.4], leapfrog_step_counts=[5], damping=0.5,
         chains=4, draws=60, seed=303),
    dict(name="drghmc_funnel11_k3", alg="drghmc", model=dict(kind="funnel", D=11), max_proposals=3,
         leapfrog_step_sizes=[0.2, 0.05, 0.0125], leapfrog_step_counts=[10, 40, 160], damping=0.1,
         chains=8, draws=60, seed=304),  # funnel dynamics are chaotic: keep the pathwise horizon short
    dict(name="drghmc_funnel101_cfg4", alg="drghmc", model=dict(kind="funnel", D=101),
         max_proposals=3, leapfrog_step_sizes=[0.2, 0.05, 0.0125],
         leapfrog_step_counts=[10, 40, 160], damping=0.1, chains=4, draws=40, seed=20242),
    # four proposal kinds: ghosts of ghosts of ghosts (the lane-set recursion three levels deep)
    dict(name="drghmc_funnel17_k4", alg="drghmc", model=dict(kind="funnel", D=17), max_proposals=4,
         leapfrog_step_sizes=[0.3, 0.1, 0.033, 0.011], leapfrog_step_counts=[3, 6, 12, 24], damping=0.3,
         chains=8, draws=40, seed=307),
    # the funnel with a diagonal metric and without probabilistic retry
    dict(name="drghmc_funnel33_k2_metric_noretry", alg="drghmc", model=dict(kind="funnel", D=33), max_proposals=2,
         leapfrog_step_sizes=[0.25, 0.08], leapfrog_step_counts=[4, 12], damping=0.5, prob_retry=False,
         metric=dict(kind="linspace", lo=0.7, hi=1.4), chains=8, draws=40, seed=308),
    # --- Metropolis / Metropolis-Hastings (bayes_kit/metropolis.py) with seeded user proposals ---
    dict(name="metropolis_rw_iso3", alg="metropolis", model=dict(kind="iso_gaussian", D=3),
         proposal=dict(kind="normal", scale=0.6, seed=9001), chains=6, draws=80, seed=501),
    dict(name="mh_ar_iso2", alg="mh", model=dict(kind="iso_gaussian", D=2),
         proposal=dict(kind="normal", a=0.8, scale=0.5, seed=9002), chains=6, draws=80, seed=502),
    dict(name="metropolis_pcg_seed", alg="metropolis", model=dict(kind="std_normal"),
         proposal=dict(kind="normal", scale=1.1, seed=9003), chains=2, draws=60, pcg_seed=77),
    dict(name="drghmc_diag40", alg="drghmc",
         model=dict(kind="diag_gaussian", D=40, log10_lo=0, log10_hi=1), max_proposals=2,
         leapfrog_step_sizes=[0.3, 0.1], leapfrog_step_counts=[3, 9], damping=0.4,
         chains=4, draws=60, seed=306),  # D >= 32: partial refresh through the wavefront-per-chain path
    dict(name="drghmc_diag16_metric", alg="drghmc",
         model=dict(kind="diag_gaussian", D=16, log10_lo=0, log10_hi=1), max_proposals=3,
         leapfrog_step_sizes=[0.5, 0.25, 0.1], leapfrog_step_counts=[3, 6, 12], damping=0.3,
         metric=dict(kind="linspace", lo=0.5, hi=1.5), chains=8, draws=80, seed=305),
    # --- round 4: the edges of the paths added in that round ---
    # D = 129: the last shape of the one-launch funnel proposal (128 coordinate rows); D = 130: the first one past it
    # (the library's funnel then sums sequentially; DRGHMC runs counted steps or host-sized launches)
    dict(name="drghmc_funnel129_k3", alg="drghmc", model=dict(kind="funnel", D=129), max_proposals=3,
         leapfrog_step_sizes=[0.15, 0.05, 0.015], leapfrog_step_counts=[4, 8, 16], damping=0.2, chains=4, draws=20, seed=309),
    dict(name="drghmc_funnel130_k2", alg="drghmc", model=dict(kind="funnel", D=130), max_proposals=2,
         leapfrog_step_sizes=[0.15, 0.05], leapfrog_step_counts=[4, 12], damping=0.2, chains=4, draws=20, seed=310),
    # a full momentum refresh every draw (damping = 1) with three stages, D >= 32
    dict(name="drghmc_iso64_k3_damp1", alg="drghmc", model=dict(kind="iso_gaussian", D=64), max_proposals=3,
         leapfrog_step_sizes=[0.9, 0.3, 0.1], leapfrog_step_counts=[2, 6, 18], damping=1.0, chains=4, draws=40, seed=311),
    # MALA at the largest D of its one-pass step kernel
    dict(name="mala_diag1024", alg="mala", model=dict(kind="diag_gaussian", D=1024, log10_lo=0, log10_hi=2),
         epsilon=2e-4, chains=4, draws=8, seed=206),
    # a single-chain PCG64 stream with D > 1 (the one-launch-per-draw path of the drop-in mode)
    dict(name="mala_pcg_d5", alg="mala", model=dict(kind="iso_gaussian", D=5), epsilon=0.15, chains=2, draws=60,
         pcg_seed=4321),
    # HMC: one leapfrog step with a non-trivial metric
    dict(name="hmc_diag40_metric_steps1", alg="hmc", model=dict(kind="diag_gaussian", D=40, log10_lo=0, log10_hi=1),
         stepsize=0.1, steps=1, chains=4, draws=40, seed=105, metric=dict(kind="linspace", lo=0.6, hi=1.3)),
]


def make_smc_model(spec):
    if spec["kind"] == "ref_binomial":
        from test.models.binomial import Binomial

        return Binomial(alpha=2, beta=3, x=5, N=15, seed=spec["init_seed"])
    if spec["kind"] == "gauss_prior_lik":
        g = np.random.default_rng(spec["data_seed"])
        D = spec["D"]
        return models.GaussPriorLik(y=g.normal(size=D) * 1.5, prec=np.logspace(0, 1.5, D), prior_scale=spec["prior_scale"])
    raise KeyError(spec)


def smc_initial(case, model):
    """Initial particles: the Binomial model's own ``initial_state`` (as test_tempered_smc.py:14-18 passes it); for the
    Gaussian model prior draws from a seeded Generator."""
    if case["model"]["kind"] == "ref_binomial":
        return model.initial_state
    g = np.random.default_rng(case["init_seed"])
    init = g.normal(size=(case["M"], model.dims())) * case["model"]["prior_scale"]
    return lambda i: init[i]


def run_smc_case(case):
    """The reference's TemperedLikelihoodSMC + metropolis_kernel (bayes_kit/smc.py:12-89) under ``np.random.seed``.
    Stored: the particles after every move and after every resampling, the ancestor indices, and the values the
    run took from the global stream in consumption order (per temperature: for every particle D standard normals and
    one uniform, then choice's M uniforms) -- recorded by wrapping ``np.random.normal / uniform / choice`` with
    pass-through recorders for the duration of the run."""
    from bayes_kit.smc import TemperedLikelihoodSMC, metropolis_kernel

    M, N, scale = case["M"], case["N"], case["scale"]
    model = make_smc_model(case["model"])
    smc = TemperedLikelihoodSMC(model, M, N, smc_initial(case, model), metropolis_kernel(scale))
    D = smc.D
    theta0 = smc.thetas.copy()
    rec = dict(z=[], u=[], cu=[], idx=[])
    orig = (np.random.normal, np.random.uniform, np.random.choice)

    def normal(loc=0.0, scale=1.0, size=None):
        out = orig[0](loc=loc, scale=scale, size=size)
        rec["z"].append((np.asarray(out) - np.asarray(loc)) / scale)  # (diagnostic only; the exact values come below)
        return out

    def uniform(*a, **k):
        out = orig[1](*a, **k)
        rec["u"].append(out)
        return out

    def choice(a, size=None, replace=True, p=None):
        st = np.random.get_state()
        out = orig[2](a, size=size, replace=replace, p=p)
        after = np.random.get_state()
        np.random.set_state(st)
        rec["cu"].append(np.random.random_sample(size))
        np.random.set_state(after)
        rec["idx"].append(np.asarray(out, dtype=np.int64))
        return out

    # the exact standard normals: the same seed replayed through a private RandomState in the same call pattern
    shadow = np.random.RandomState(case["seed"])
    z_exact = np.empty((N, M, D))
    np.random.seed(case["seed"])
    moved, after = np.empty((N, M, D)), np.empty((N, M, D))
    np.random.normal, np.random.uniform, np.random.choice = normal, uniform, choice
    try:
        for n in range(1, N + 1):
            pre = smc.thetas.copy()
            # transition() overwrites thetas in place and then rebinds: capture the moved particles through choice()
            smc.transition(n)
            after[n - 1] = smc.thetas
            for m in range(M):
                z_exact[n - 1, m] = shadow.standard_normal(size=D)
                assert shadow.uniform() == rec["u"][(n - 1) * M + m]
            assert np.array_equal(shadow.random_sample(M), rec["cu"][n - 1])
            del pre
    finally:
        np.random.normal, np.random.uniform, np.random.choice = orig
    u = np.asarray(rec["u"]).reshape(N, M)
    cu = np.stack(rec["cu"])
    idx = np.stack(rec["idx"])
    # the moved particles are not observable from outside transition(); recompute them as the reference's kernel
    # does from the stored values and check them through the resampling: after[n] == moved[n][idx[n]]
    cur = theta0.copy()
    for n in range(N):
        t = n / N
        for m in range(M):
            th = np.atleast_1d(cur[m])
            prop = th + scale * z_exact[n, m]
            lp = lambda x: model.log_likelihood(x) * t + model.log_prior(x)  # noqa: E731
            cur[m] = prop if np.log(u[n, m]) < lp(prop) - lp(th) else th
        moved[n] = cur
        assert np.array_equal(cur[idx[n]], after[n]), n
        cur = after[n].copy()
    final_state = np.random.get_state(legacy=False)
    # (the particles after resampling are moved[n][idx[n]] -- checked above -- and are not stored twice)
    return dict(case=np.array(json.dumps(case)), theta0=theta0, moved=moved, idx=idx,
                normals=z_exact, uniforms=u, choice_uniforms=cu,
                final_pos=np.int64(final_state["state"]["pos"]), final_key=final_state["state"]["key"][:8].copy(),
                final_has_gauss=np.int64(final_state["has_gauss"]))


SMC_CASES = [
    # the reference's own test (test/test_tempered_smc.py:8-30): Binomial model, M = 75, N = 15, scale 0.5
    dict(name="smc_ref_binomial", model=dict(kind="ref_binomial", init_seed=11), M=75, N=15, scale=0.5, seed=20245),
    # elementwise Gaussian prior / likelihood; D odd, so the polar method's cached second normal crosses particles
    dict(name="smc_gauss5_m512", model=dict(kind="gauss_prior_lik", D=5, data_seed=3, prior_scale=2.0), M=512, N=8,
         scale=0.35, seed=20246, init_seed=7),
    dict(name="smc_gauss3_m2048", model=dict(kind="gauss_prior_lik", D=3, data_seed=4, prior_scale=1.5), M=2048, N=3,
         scale=0.25, seed=20247, init_seed=8),
]


def ar1(rng, n, phi):
    x = np.empty(n)
    x[0] = rng.normal()
    for i in range(1, n):
        x[i] = phi * x[i - 1] + rng.normal()
    return x


def run_diagnostics():
    rng = np.random.default_rng(777)
    out = {}
    # rhat on 64 chains x 200 draws of mildly different locations (bayes_kit/rhat.py:111-171)
    chains = [rng.normal(loc=0.05 * rng.normal(), size=200) for _ in range(64)]
    out["rhat_chains"] = np.stack(chains)
    out["rhat"] = np.float64(ref.rhat(chains))
    out["split_rhat"] = np.float64(ref.rhat.__globals__["split_rhat"](chains))
    # ragged chains (allowed by rhat.py:163-171)
    lens = [50, 61, 73, 40, 97]
    ragged = [rng.normal(size=n) for n in lens]
    out["ragged_lens"] = np.asarray(lens)
    out["ragged_flat"] = np.concatenate(ragged)
    out["ragged_rhat"] = np.float64(ref.rhat(ragged))
    out["ragged_split_rhat"] = np.float64(ref.rhat.__globals__["split_rhat"](ragged))
    # rank-normalised rhat, small (scalar ppf is slow in the reference)
    rk = [rng.standard_cauchy(size=100) for _ in range(8)]
    out["rank_chains"] = np.stack(rk)
    out["rank_normalized_rhat"] = np.float64(ref.rhat.__globals__["rank_normalized_rhat"](rk))
    out["rank_normalized"] = np.asarray(ref.rhat.__globals__["rank_normalize_chains"](rk))
    # ESS / IAT / autocorr on 16 AR(1) chains x 1000 (ess.py, iat.py, autocorr.py)
    phis = np.linspace(-0.6, 0.9, 16)
    ar = np.stack([ar1(rng, 1000, p) for p in phis])
    out["ar_chains"] = ar
    out["ar_phi"] = phis
    out["ar_autocorr"] = np.stack([ref.autocorr(c) for c in ar])
    out["ar_ess"] = np.asarray([ref.ess(c) for c in ar])
    out["ar_ess_ipse"] = np.asarray([ref.ess_ipse(c) for c in ar])
    out["ar_ess_imse"] = np.asarray([ref.ess_imse(c) for c in ar])
    out["ar_iat"] = np.asarray([ref.iat(c) for c in ar])
    out["ar_iat_ipse"] = np.asarray([ref.iat_ipse(c) for c in ar])
    # odd / short lengths
    short = [rng.normal(size=n) for n in (4, 5, 7, 33)]
    out["short_lens"] = np.asarray([4, 5, 7, 33])
    out["short_flat"] = np.concatenate(short)
    out["short_ess"] = np.asarray([ref.ess(c) for c in short])
    out["short_autocorr_flat"] = np.concatenate([ref.autocorr(c) for c in short])
    # round 4: LONG chains (20,000 draws: the library's own FFT path, from 16,384 draws on).  The series are regenerated
    # from (long_seed, long_phi) by tests/helpers.long_ar_chains -- only the reference's OUTPUTS are stored
    from tests.helpers import long_ar_chains

    out["long_seed"], out["long_n"] = np.int64(4242), np.int64(20000)
    out["long_phi"] = np.asarray([-0.5, 0.3, 0.9, 0.99])
    lc = long_ar_chains(int(out["long_seed"]), int(out["long_n"]), out["long_phi"])
    out["long_ess"] = np.asarray([ref.ess(c) for c in lc])
    out["long_ess_ipse"] = np.asarray([ref.ess_ipse(c) for c in lc])
    out["long_iat"] = np.asarray([ref.iat(c) for c in lc])
    out["long_autocorr_head"] = np.stack([ref.autocorr(c)[:64] for c in lc])
    out["long_autocorr_tail"] = np.stack([ref.autocorr(c)[-8:] for c in lc])
    # pooled ranks with MANY TIES (draws of small integers): how the reference ranks equal values (rhat.py:27-69)
    tg = np.random.default_rng(99)
    ties = [tg.integers(0, 7, size=80).astype(np.float64) for _ in range(6)]
    out["ties_chains"] = np.stack(ties)
    out["ties_ranks"] = np.asarray(ref.rhat.__globals__["rank_chains"](ties))
    out["ties_rank_normalized"] = np.asarray(ref.rhat.__globals__["rank_normalize_chains"](ties))
    out["ties_rank_normalized_rhat"] = np.float64(ref.rhat.__globals__["rank_normalized_rhat"](ties))
    return out


def main():
    only = set(sys.argv[1:])  # optional: regenerate just the named cases
    for case in SAMPLER_CASES:
        if only and case["name"] not in only:
            continue
        res = run_sampler_case(case)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **res)
        print("%-28s draws %s  mean grad calls/draw %.1f  size %d B"
              % (case["name"], res["draws"].shape, res["grad_calls"].mean(), os.path.getsize(path)))
    for case in SMC_CASES:
        if only and case["name"] not in only:
            continue
        res = run_smc_case(case)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **res)
        print("%-28s particles %s  distinct ancestors at the end %d  size %d B"
              % (case["name"], res["moved"].shape, len(np.unique(res["idx"][-1])), os.path.getsize(path)))
    if only and "diagnostics" not in only:
        return
    d = run_diagnostics()
    path = os.path.join(HERE, "diagnostics.npz")
    np.savez_compressed(path, **d)
    print("diagnostics", os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
