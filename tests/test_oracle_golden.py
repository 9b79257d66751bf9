"""The oracle restatement vs vectors produced by the real reference (CPU, no GPU).

Bar: theta, returned logp, per-draw gradient-call counts and the bit generator's final
state are all BIT-EXACT (the oracle uses the same NumPy operations in the same order as
bayes_kit/hmc.py, mala.py, drghmc.py).
"""
import numpy as np
import pytest

from oracle import diagnostics as od
from tests.helpers import SAMPLER_CASES, load_case, oracle_model, oracle_sampler, rng_state_words


class _Counting:
    def __init__(self, inner):
        self._inner, self.n = inner, 0

    def dims(self):
        return self._inner.dims()

    def log_density(self, t):
        return self._inner.log_density(t)

    def log_density_gradient(self, t):
        self.n += 1
        return self._inner.log_density_gradient(t)


@pytest.mark.parametrize("name", SAMPLER_CASES)
def test_sampler_matches_reference_bit_for_bit(name):
    case, z = load_case(name)
    N, C, D = z["draws"].shape
    for c in range(C):
        model = _Counting(oracle_model(case["model"]))
        s = oracle_sampler(case, c, model=model)
        extra = model.n  # MALA evaluates the gradient once in its constructor
        np.testing.assert_array_equal(np.asarray(s._theta, dtype=np.float64), z["theta0"][c])
        for n in range(N):
            before = model.n
            th, lp = s.sample()
            assert np.array_equal(th, z["draws"][n, c]), (name, c, n)
            assert lp == z["logp"][n, c], (name, c, n)
            calls = model.n - before
            if case["alg"] == "drghmc":
                # the reference seeds its cache with one extra call on the first draw only
                assert calls == z["grad_calls"][n, c], (name, c, n)
            elif case["alg"] == "hmc":
                assert calls == case["steps"] + 1
            elif case["alg"] == "mala":
                assert calls == 1
            else:
                assert calls == 0  # Metropolis(-Hastings) never asks for a gradient
        np.testing.assert_array_equal(rng_state_words(s._rng), z["rng_state"][c])
        if case["alg"] == "drghmc":
            np.testing.assert_array_equal(s._rho, z["rho_final"][c])
        del extra


def test_hmc_one_step_equals_mala():
    # known-answer equivalence pinned by the reference: test/test_equivalencies.py:12-32
    from oracle.models import StdNormal
    from oracle.samplers import HMCDiag, MALA

    init = np.array([0.2])
    eps = 0.02
    hmc = HMCDiag(StdNormal(), stepsize=eps, steps=1, init=init, seed=123)
    mala = MALA(StdNormal(), epsilon=0.5 * eps**2, init=init, seed=123)
    d1 = np.array([hmc.sample()[0] for _ in range(50)])
    d2 = np.array([mala.sample()[0] for _ in range(50)])
    np.testing.assert_array_almost_equal(d1, d2)
    assert len(np.unique(d1)) > 20


def test_drghmc_schedule_tags():
    case, z = load_case("drghmc_funnel11_k3")
    s = oracle_sampler(case, 0)
    L = case["leapfrog_step_counts"]
    seen = set()
    for n in range(z["draws"].shape[0]):
        s.sample()
        tags = tuple(s.last_schedule)
        seen.add(tags)
        # gradient evaluations = sum of the step counts of the trajectories run (+1 on draw 0)
        steps = sum(L[int(t[1])] for t in tags)
        assert s.last_grad_evals == steps + (1 if n == 0 else 0)
        # slot order is a prefix-closed subsequence of the fixed 7-slot schedule
        full = ["P0", "P1", "G0(P1)", "P2", "G0(P2)", "G1(P2)", "G0(G1(P2))"]
        it = iter(full)
        assert all(t in it for t in tags), tags
    assert ("P0",) in seen and len(seen) >= 2


# ---- diagnostics ----------------------------------------------------------------
def test_diagnostics_match_reference():
    import os
    from tests.helpers import GOLDEN

    z = np.load(os.path.join(GOLDEN, "diagnostics.npz"))
    chains = list(z["rhat_chains"])
    assert od.rhat(chains) == z["rhat"]
    assert od.split_rhat(chains) == z["split_rhat"]
    lens = z["ragged_lens"]
    ragged = np.split(z["ragged_flat"], np.cumsum(lens)[:-1])
    assert od.rhat(ragged) == z["ragged_rhat"]
    assert od.split_rhat(ragged) == z["ragged_split_rhat"]
    rk = list(z["rank_chains"])
    assert od.rank_normalized_rhat(rk) == z["rank_normalized_rhat"]
    np.testing.assert_array_equal(np.asarray(od.rank_normalize_chains(rk)), z["rank_normalized"])
    for i, ch in enumerate(z["ar_chains"]):
        np.testing.assert_array_equal(od.autocorr(ch), z["ar_autocorr"][i])
        assert od.ess(ch) == z["ar_ess"][i]
        assert od.ess_ipse(ch) == z["ar_ess_ipse"][i]
        assert od.ess_imse(ch) == z["ar_ess_imse"][i]
        assert od.iat(ch) == z["ar_iat"][i]
        assert od.iat_ipse(ch) == z["ar_iat_ipse"][i]
    short = np.split(z["short_flat"], np.cumsum(z["short_lens"])[:-1])
    np.testing.assert_array_equal(np.asarray([od.ess(c) for c in short]), z["short_ess"])
    np.testing.assert_array_equal(np.concatenate([od.autocorr(c) for c in short]), z["short_autocorr_flat"])
    tc = list(z["ties_chains"])   # many ties: small integers
    np.testing.assert_array_equal(np.asarray(od.rank_chains(tc)), z["ties_ranks"])
    np.testing.assert_array_equal(np.asarray(od.rank_normalize_chains(tc)), z["ties_rank_normalized"])
    assert od.rank_normalized_rhat(tc) == z["ties_rank_normalized_rhat"]
    # long chains (20,000 draws), regenerated from their seed
    from tests.helpers import long_ar_chains

    for i, ch in enumerate(long_ar_chains(z["long_seed"], z["long_n"], z["long_phi"])):
        assert od.ess(ch) == z["long_ess"][i] and od.ess_ipse(ch) == z["long_ess_ipse"][i] and od.iat(ch) == z["long_iat"][i]
        ac = od.autocorr(ch)
        np.testing.assert_array_equal(ac[:64], z["long_autocorr_head"][i])
        np.testing.assert_array_equal(ac[-8:], z["long_autocorr_tail"][i])


def test_diagnostics_known_answers_from_reference_tests():
    # literal vectors held by the reference's own tests
    # test/test_autocorr.py:10-14
    np.testing.assert_allclose(od.autocorr([1, 0, 0, 0]), [1.0, -0.083, -0.167, -0.25], atol=0.001)
    # test/test_iat.py:72-80
    assert od._end_pos_pairs([]) == 0
    assert od._end_pos_pairs([1]) == 0
    assert od._end_pos_pairs([1, 0.4]) == 2
    assert od._end_pos_pairs([1, -0.4]) == 2
    assert od._end_pos_pairs([1, -0.5, 0.25, -0.3]) == 2
    assert od._end_pos_pairs([1, -0.5, 0.25, -0.1]) == 4
    assert od._end_pos_pairs([1, -0.5, 0.25, -0.3, 0.05]) == 2
    assert od._end_pos_pairs([1, -0.5, 0.25, -0.1, 0.05]) == 4
    # test/test_rhat.py:71-79
    got = od.split_chains([[1, 2, 3], [4, 5, 6, 7]])
    assert [list(g) for g in got] == [[1, 2], [3], [4, 5], [6, 7]]
    # test/test_rhat.py:113-126
    got = od.rank_chains([[4.2, 5.7], [7.2, 6.1], [-12.9, 107]])
    assert [list(g) for g in got] == [[2, 3], [5, 4], [1, 6]]
    # rhat.py:86-87 docstring example (0.325 offset)
    got = od.rank_normalize_chains([[4.2, 5.7], [7.2, 6.1], [-12.9, 107]])
    import scipy.stats as st
    want = [[st.norm.ppf((r - 0.325) / (6 - 0.25)) for r in row] for row in [[2, 3], [5, 4], [1, 6]]]
    np.testing.assert_allclose(got, want, rtol=0, atol=0)
    # brute-force BDA3 R-hat (test/test_rhat.py:19-52 idea) for equal-length chains
    rng = np.random.default_rng(5)
    chains = [rng.normal(size=100) for _ in range(4)]
    N = 100
    psij = np.array([c.mean() for c in chains])
    B = N * psij.var(ddof=1)
    W = np.mean([c.var(ddof=1) for c in chains])
    expect = np.sqrt(((N - 1) / N * W + B / N) / W)
    np.testing.assert_allclose(od.rhat(chains), expect, rtol=1e-12)
    # error behaviour: rhat.py:159-162, ess.py:67-68, autocorr.py:23-24
    with pytest.raises(ValueError):
        od.rhat([[1.0, 2.0]])
    with pytest.raises(ValueError):
        od.rhat([[1.0, 2.0], [1.0]])
    with pytest.raises(ValueError):
        od.ess([1.0, 2.0, 3.0])
    with pytest.raises(ValueError):
        od.autocorr([1.0])


# ---- likelihood-tempered SMC (bayes_kit/smc.py:12-89 under np.random.seed) ----------------------------------------
def _oracle_smc(case, z, stream):
    from oracle import smc as osmc
    from tests.helpers import smc_model

    model = smc_model(case["model"])
    return osmc.TemperedLikelihoodSMC(model, case["M"], case["N"], lambda i: z["theta0"][i],
                                      osmc.metropolis_kernel(case["scale"], stream), stream)


@pytest.mark.parametrize("source", ["restated_mt19937", "numpy_randomstate", "replay"])
@pytest.mark.parametrize("name", ["smc_ref_binomial", "smc_gauss5_m512", "smc_gauss3_m2048"])
def test_smc_matches_reference_bit_for_bit(name, source):
    from oracle import smc as osmc
    from oracle.rng import LegacyStream
    from tests.helpers import smc_expected_thetas

    case, z = load_case(name)
    if source == "restated_mt19937":
        stream = LegacyStream(case["seed"])
    elif source == "numpy_randomstate":
        stream = osmc.NumpyLegacySource(np.random.RandomState(case["seed"]))
    else:
        stream = osmc.ReplaySource(z["normals"], z["uniforms"], z["choice_uniforms"])
    smc = _oracle_smc(case, z, stream)
    want = smc_expected_thetas(z)
    for n in range(1, case["N"] + 1):
        smc.transition(n)
        assert np.array_equal(smc.moved, z["moved"][n - 1]), (name, n)
        assert np.array_equal(smc.idxs, z["idx"][n - 1]), (name, n)
        assert np.array_equal(smc.thetas, want[n - 1]), (name, n)
    if source == "restated_mt19937":
        st = stream.state()
        assert st["pos"] == int(z["final_pos"]) and st["has_gauss"] == int(z["final_has_gauss"])
        assert np.array_equal(st["key"][:8], z["final_key"])


def test_smc_fixture_reproduces_the_reference_test_moments():
    # test/test_tempered_smc.py:8-30 asserts the Binomial posterior's moments at M = 75; the fixture is that run
    from scipy import stats
    from scipy.special import expit

    from tests.helpers import smc_expected_thetas

    case, z = load_case("smc_ref_binomial")
    draws = expit(smc_expected_thetas(z)[-1])
    post = stats.beta(2 + 5, 3 + 15 - 5)
    assert abs(draws.mean() - post.mean()) < 0.05 and abs(draws.var(ddof=1) - post.var()) < 0.01


@pytest.mark.parametrize("name", ["drghmc_funnel11_k3", "drghmc_funnel101_cfg4", "drghmc_funnel17_k4",
                                  "drghmc_funnel33_k2_metric_noretry", "drghmc_funnel129_k3", "drghmc_funnel130_k2"])
def test_canonical_order_funnel_oracle_vs_reference_golden(name):
    """oracle.models.FunnelCanonical (sum x^2 in the HIP library's 16-class order) against the reference's goldens (np.dot):
    the ONLY difference is the summation order, which the funnel's chaotic flow amplifies -- this is where the widening
    bound of tests/sampler_parity.funnel_tol belongs (CPU vs CPU); every accept / retry decision and the final stream
    state are exact.  The GPU is held to the flat 1e-9 against THIS oracle (tests/test_gpu_samplers.py)."""
    from oracle import models as om
    from tests.sampler_parity import funnel_tol

    case, z = load_case(name)
    N, C, D = z["draws"].shape
    for c in range(C):
        s = oracle_sampler(case, c, model=om.FunnelCanonical(D))
        for n in range(N):
            th, lp = s.sample()
            np.testing.assert_allclose(th, z["draws"][n, c], **funnel_tol(n))
            np.testing.assert_allclose(lp, z["logp"][n, c], **funnel_tol(n))
        np.testing.assert_array_equal(rng_state_words(s._rng), z["rng_state"][c])
