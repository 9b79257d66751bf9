"""Shared test helpers: load golden cases and rebuild their oracle-side objects."""
import json
import os

import numpy as np

from oracle import models as omodels
from oracle import samplers as osamplers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SAMPLER_CASES = [
    "hmc_stdnormal", "hmc_steps0", "hmc_iso4", "hmc_iso128_cfg2", "hmc_diag16_metric",
    "hmc_diag1024_cfg3", "hmc_pcg_seed",
    "mala_readme_cfg1", "mala_stdnormal", "mala_iso8", "mala_diag16", "mala_diag48", "mala_init",
    "drghmc_stdnormal_k3", "drghmc_iso4_k2_noretry", "drghmc_k1", "drghmc_funnel11_k3",
    "drghmc_funnel101_cfg4", "drghmc_diag16_metric", "drghmc_diag40", "drghmc_funnel17_k4",
    "drghmc_funnel33_k2_metric_noretry",
    "hmc_ref_binomial", "mala_ref_binomial", "drghmc_ref_binomial",
    "metropolis_rw_iso3", "mh_ar_iso2", "metropolis_pcg_seed",
    "drghmc_funnel129_k3", "drghmc_funnel130_k2", "drghmc_iso64_k3_damp1", "mala_diag1024", "mala_pcg_d5",
    "hmc_diag40_metric_steps1",
]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = json.loads(str(z["case"]))
    return case, z


def oracle_model(spec):
    kind = spec["kind"]
    if kind == "std_normal":
        return omodels.StdNormal()
    if kind == "iso_gaussian":
        return omodels.IsoGaussian(spec["D"])
    if kind == "diag_gaussian":
        return omodels.DiagGaussian(np.logspace(spec["log10_lo"], spec["log10_hi"], spec["D"]))
    if kind == "funnel":
        return omodels.Funnel(spec["D"])
    if kind == "ref_binomial":
        from tests.host_models import Binomial

        return Binomial(alpha=2, beta=3, x=5, N=15)
    raise KeyError(kind)


def case_metric(case, D):
    spec = case.get("metric")
    if spec is None:
        return None
    if spec["kind"] == "linspace":
        return np.linspace(spec["lo"], spec["hi"], D)
    return np.ones(D)


def case_seed(case, c):
    if "pcg_seed" in case:
        return case["pcg_seed"] + c
    return np.random.Philox(key=[case["seed"], c])


def host_proposal(spec, c):
    """The golden cases' user-side proposal callbacks for chain c (NumPy, single chain):
    theta* ~ N(a*theta, scale^2 I) from the proposal's own Philox(key=[seed, c]) stream, and its
    log transition density up to a constant."""
    prng = np.random.Generator(np.random.Philox(key=[spec["seed"], c]))
    a, scale = spec.get("a", 1.0), spec["scale"]

    def proposal_fn(theta):
        return prng.normal(loc=a * theta, scale=scale)

    def transition_lp_fn(to, frm):
        r = (to - a * frm) / scale
        return -0.5 * np.sum(r * r)

    return proposal_fn, transition_lp_fn


def oracle_sampler(case, c, model=None):
    """Oracle sampler for chain c of a golden case."""
    model = model or oracle_model(case["model"])
    D = model.dims()
    init = None if case.get("init") is None else np.asarray(case["init"], dtype=np.float64).copy()
    metric = case_metric(case, D)
    seed = case_seed(case, c)
    alg = case["alg"]
    if alg == "hmc":
        return osamplers.HMCDiag(model, case["stepsize"], case["steps"], metric_diag=metric, init=init, seed=seed)
    if alg == "mala":
        return osamplers.MALA(model, case["epsilon"], init=init, seed=seed)
    if alg == "drghmc":
        return osamplers.DrGhmcDiag(
            model, case["max_proposals"], case["leapfrog_step_sizes"], case["leapfrog_step_counts"],
            case["damping"], metric_diag=metric, init=init, seed=seed, prob_retry=case.get("prob_retry", True))
    if alg in ("metropolis", "mh"):
        proposal_fn, transition_lp_fn = host_proposal(case["proposal"], c)
        if alg == "metropolis":
            return osamplers.Metropolis(model, proposal_fn, init=init, seed=seed)
        return osamplers.MetropolisHastings(model, proposal_fn, transition_lp_fn, init=init, seed=seed)
    raise KeyError(alg)


def rng_state_words(gen):
    st = gen.bit_generator.state
    if st["bit_generator"] == "Philox":
        return np.concatenate([
            np.asarray(st["state"]["key"], dtype=np.uint64),
            np.asarray(st["state"]["counter"], dtype=np.uint64),
            np.asarray(st["buffer"], dtype=np.uint64),
            np.asarray([st["buffer_pos"]], dtype=np.uint64)])
    s = st["state"]
    m = 2**64 - 1
    return np.asarray([s["state"] >> 64, s["state"] & m, s["inc"] >> 64, s["inc"] & m], dtype=np.uint64)


def long_ar_chains(seed, n, phis):
    """AR(1) series x[i] = phi x[i-1] + e[i] from a seeded PCG64 stream (deterministic everywhere): the inputs of the
    long-chain diagnostics fixture, regenerated instead of stored (4 x 20,000 doubles)."""
    rng = np.random.default_rng(int(seed))
    out = []
    for phi in phis:
        e = rng.normal(size=int(n))
        x = np.empty(int(n))
        x[0] = e[0]
        for i in range(1, int(n)):
            x[i] = phi * x[i - 1] + e[i]
        out.append(x)
    return out


SMC_CASES = ["smc_ref_binomial", "smc_gauss5_m512", "smc_gauss3_m2048"]


def smc_model(spec):
    """The model of an SMC fixture (tests/golden/make_golden.py::make_smc_model), rebuilt without the reference."""
    if spec["kind"] == "ref_binomial":
        from tests.host_models import Binomial

        return Binomial(alpha=2, beta=3, x=5, N=15)
    if spec["kind"] == "gauss_prior_lik":
        g = np.random.default_rng(spec["data_seed"])
        D = spec["D"]
        return omodels.GaussPriorLik(y=g.normal(size=D) * 1.5, prec=np.logspace(0, 1.5, D), prior_scale=spec["prior_scale"])
    raise KeyError(spec)


def smc_expected_thetas(z):
    """The particles after every resampling: moved[n][idx[n]] (the generator checked this against the reference)."""
    return np.stack([z["moved"][n][z["idx"][n]] for n in range(z["idx"].shape[0])])
