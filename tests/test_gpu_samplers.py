"""Parity of the HIP samplers with the reference's golden vectors and with the oracle."""
import os

import numpy as np
import pytest
import torch

import bayes_kit_amd as bk
from tests.sampler_parity import check_checkpoint_resume, check_many_chain, check_single_chain_host_model, funnel_tol

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

MANY = ["hmc_stdnormal", "hmc_steps0", "hmc_iso4", "hmc_iso128_cfg2", "hmc_diag16_metric", "hmc_diag1024_cfg3",
        "mala_stdnormal", "mala_iso8", "mala_diag16", "mala_diag48", "mala_init",
        "drghmc_stdnormal_k3", "drghmc_iso4_k2_noretry", "drghmc_k1", "drghmc_funnel11_k3",
        "drghmc_funnel101_cfg4", "drghmc_diag16_metric", "drghmc_diag40", "drghmc_funnel17_k4",
        "drghmc_funnel33_k2_metric_noretry", "metropolis_rw_iso3", "mh_ar_iso2",
        # round 4: the edges of the new paths (one-launch proposal at its last D and one past it, full refresh, MALA's
        # step kernel at its largest D, one leapfrog step with a metric)
        "drghmc_funnel129_k3", "drghmc_funnel130_k2", "drghmc_iso64_k3_damp1", "mala_diag1024", "hmc_diag40_metric_steps1"]


@pytest.fixture(scope="module")
def ops():
    return bk._lib.default_ops()


@pytest.mark.parametrize("name", MANY)
def test_many_chain_vs_reference_golden(name, ops):
    check_many_chain(name, ops)


@pytest.mark.parametrize("name", [n for n in MANY if n.startswith("drghmc")])
def test_drghmc_model_opaque_device_counts_vs_reference_golden(name, ops):
    """The reference's DRGHMC fixtures through the model-opaque path with lane counts on the device: one counted
    gradient op + one counted kick+drift launch per leapfrog step (drghmc.py:280-283), no host read, one hipGraph."""
    s = check_many_chain(name, ops, path="step", device_counts=True)
    assert s._dev_counts and not s._one_launch and s._use_graph and s.host_syncs_per_draw == 0


@pytest.mark.parametrize("name", [n for n in MANY if n.startswith("hmc")])
def test_hmc_step_by_step_path_vs_reference_golden(name, ops):
    """The model-opaque path (one kick+drift launch and one gradient op per leapfrog step),
    which is what bench.py measures; the default for built-in Gaussians is the fused one."""
    s = check_many_chain(name, ops, path="step")
    assert not s._fused


@pytest.mark.parametrize("name", ["hmc_pcg_seed", "hmc_iso4", "mala_stdnormal", "mala_init",
                                  "drghmc_stdnormal_k3", "drghmc_k1",
                                  # the reference's own scipy-based test model with its finite-difference gradient
                                  "hmc_ref_binomial", "mala_ref_binomial", "drghmc_ref_binomial",
                                  "metropolis_rw_iso3", "mh_ar_iso2", "metropolis_pcg_seed", "mala_pcg_d5", "mala_diag16"])
def test_single_chain_drop_in_vs_reference_golden(name, ops):
    check_single_chain_host_model(name, ops, chains=[0, 1])


def test_readme_example_cfg1(ops):
    # BASELINE.json config 1 (README.md:13-32): MALA, StdNormal, int seed, 1000 draws
    check_single_chain_host_model("mala_readme_cfg1", ops)


def test_torch_autograd_model_matches_builtin(ops):
    lam = torch.logspace(0, 1, 16, dtype=torch.float64, device=ops.device)
    tm = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), 16)
    a = bk.HMCDiag(tm, 0.05, 8, chains=300, seed=9)
    b = bk.HMCDiag(bk.DiagGaussian(lam.cpu()), 0.05, 8, chains=300, seed=9)
    for _ in range(10):
        ta, la = a.sample()
        tb, lb = b.sample()
        np.testing.assert_allclose(ta.cpu().numpy(), tb.cpu().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-11)


@pytest.mark.parametrize("two_pass", [True, False])
def test_mala_padded_row_pitch_gives_the_same_draws(ops, monkeypatch, two_pass):
    """From 4,096 chains on, MALA keeps its [D, C] arrays with the rows 144 columns further apart than C (rows of a chain
    off one memory channel); the draws, log densities and stream positions are those of dense arrays."""
    lam = np.logspace(0, 1, 40)
    a = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=4096, seed=3, two_pass=two_pass)
    monkeypatch.setenv("BK_STATE_PAD", "0")
    b = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=4096, seed=3, two_pass=two_pass)
    monkeypatch.delenv("BK_STATE_PAD")
    assert a._state_pad == 144 and b._state_pad == 0 and a._theta_dc.stride(0) == 4096 + 144
    for _ in range(6):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert tuple(ta.shape) == (4096, 40) and torch.equal(ta, tb) and torch.equal(la, lb)
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    sd = a.state_dict()
    c = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=4096, seed=99, two_pass=two_pass)
    c.load_state_dict(sd)
    assert torch.equal(c.sample()[0], a.sample()[0])


def test_torch_model_written_for_the_engine_layout(ops):
    """TorchModel(layout="dc"): the user function takes the (D, C) array itself -- same draws as the (C, D) form and as
    the built-in target, and the gradient comes back chain-contiguous (the streamed kick + drift, no LDS turn)."""
    lam = torch.logspace(0, 1, 16, dtype=torch.float64, device=ops.device)
    cd = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), 16)
    dc = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), 16, layout="dc")
    for mk in (lambda m: bk.HMCDiag(m, 0.05, 8, chains=300, seed=9), lambda m: bk.MALA(m, 0.01, chains=300, seed=9),
               lambda m: bk.DrGhmcDiag(m, 2, [0.1, 0.03], [4, 8], 0.3, chains=300, seed=9)):
        a, b, c = mk(cd), mk(dc), mk(bk.DiagGaussian(lam.cpu()))
        for _ in range(6):
            ta, la = a.sample()
            tb, lb = b.sample()
            tc, lc = c.sample()
            np.testing.assert_allclose(tb.cpu().numpy(), ta.cpu().numpy(), rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(tb.cpu().numpy(), tc.cpu().numpy(), rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(lb.cpu().numpy(), lc.cpu().numpy(), rtol=1e-11)
    th = torch.randn((16, 300), dtype=torch.float64, device=ops.device)
    _, g = dc.log_density_gradient(th.t())
    assert tuple(g.shape) == (300, 16) and g.stride(0) == 1   # (C, D) view of a chain-contiguous (D, C) array
    with pytest.raises(ValueError):
        bk.TorchModel(lambda Th: Th.sum(dim=1), 3, layout="rows")


def test_row_major_model_output_goes_through_lds_transpose(ops):
    # a model that returns a fresh row-major (C, D) gradient (dimension-contiguous)
    class RowMajor:
        batched = True

        def dims(self):
            return 64

        def log_density(self, Th):
            return -0.5 * (Th * Th).sum(dim=1)

        def log_density_gradient(self, Th):
            return -0.5 * (Th * Th).sum(dim=1), (-Th).contiguous()

    a = bk.HMCDiag(RowMajor(), 0.1, 5, chains=256, seed=3)
    b = bk.HMCDiag(bk.IsoGaussian(64), 0.1, 5, chains=256, seed=3)
    for _ in range(5):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb)
        np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-12)


def test_chain_identity_is_independent_of_batch(ops):
    # sharding invariance: chain g behaves identically whatever batch it is in
    lam = np.logspace(0, 2, 32)
    big = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 6, chains=1000, seed=77)
    part = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 6, chains=100, seed=77, chain_id0=640)
    for _ in range(4):
        tb, lb = big.sample()
        tp, lp = part.sample()
        assert torch.equal(tb[640:740], tp)
        assert torch.equal(lb[640:740], lp)


def test_full_size_cfg3_properties(ops):
    """BASELINE.json config 3 at full size (65,536 chains x D=1024, L=64): size-independent
    properties -- a scattered subset of chains reproduces a small run bit for bit (which
    tests/golden pins to the reference), energy errors are small, accept rate is sane."""
    from bench import make_cfg3_sampler

    C = 65536
    s = make_cfg3_sampler(C, 0, ops.device)
    small = make_cfg3_sampler(64, 40000, ops.device)
    for _ in range(2):
        tb, lb = s.sample()
        ts, ls = small.sample()
        assert torch.equal(tb[40000:40064], ts)
        assert torch.equal(lb[40000:40064], ls)
    rate = s.accept_rate()
    assert 0.6 < rate < 0.99, rate
    assert torch.isfinite(tb).all() and torch.isfinite(lb).all()


def test_drghmc_many_chains_match_oracle_per_chain(ops):
    """2,048 funnel chains in lockstep (compaction, ghost levels) vs the oracle chain by chain
    on a scattered subset: decisions, theta and the RNG position all agree."""
    from oracle import models as om
    from oracle import samplers as osamp

    D, C, seed, N = 11, 2048, 909, 12
    args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)
    s = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=C, seed=seed)
    draws = []
    stages = set()
    for _ in range(N):
        th, lp = s.sample()
        draws.append((th.cpu().numpy(), lp.cpu().numpy()))
        stages.update(t for t, _ in s.last_stage_lanes)
    assert {"P0", "P1", "G0(P1)"} <= stages
    st = s.rng_state()
    for c in list(range(0, C, 97)) + [C - 1]:
        o = osamp.DrGhmcDiag(om.Funnel(D), *args, seed=np.random.Philox(key=[seed, c]))
        for n in range(N):
            oth, olp = o.sample()
            np.testing.assert_allclose(draws[n][0][c], oth, **funnel_tol(n))  # 1e-9: N < 20 draws
            np.testing.assert_allclose(draws[n][1][c], olp, **funnel_tol(n))
        from tests.helpers import rng_state_words

        np.testing.assert_array_equal(st[:, c], rng_state_words(o._rng))


def test_drghmc_moments_std_normal(ops):
    # distributional check in the spirit of test/test_drghmc.py:97-117, many chains at once
    s = bk.DrGhmcDiag(bk.IsoGaussian(1), 3, [0.9, 0.45, 0.225], [2, 4, 8], 0.2, chains=4096, seed=1)
    acc = []
    for n in range(60):
        th, _ = s.sample()
        if n >= 20:
            acc.append(th[:, 0].clone())
    x = torch.stack(acc).cpu().numpy()
    assert abs(x.mean()) < 0.02 and abs(x.var() - 1.0) < 0.03


def test_diagnostics_vs_reference_golden(ops):
    """rhat / split_rhat / ess / iat / autocorr on the device vs values computed by the
    reference itself (tests/golden/diagnostics.npz).  The device gets autocorrelations by
    direct summation, the reference by FFT: rel 1e-9 on ESS/IAT, abs 1e-12 on autocorr."""
    from tests.diag_parity import check_diagnostics

    check_diagnostics(ops, ess_rtol=1e-9)


def test_cache_tiling_does_not_change_results(ops):
    lam = np.logspace(0, 2, 48)
    a = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 7, chains=1000, seed=5, chain_tile=0)
    b = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 7, chains=1000, seed=5, chain_tile=128)
    assert a._chain_tile == 1000 and b._chain_tile == 128
    for _ in range(4):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb)


def test_full_size_cfg4_properties(ops):
    """BASELINE.json config 4 at per-GPU size (32,768 funnel chains, D=101, K=3): a scattered
    block of chains reproduces a small run exactly (sharding invariance through compaction,
    ghost levels and scatter), lane accounting is consistent, everything stays finite."""
    args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)
    big = bk.DrGhmcDiag(bk.Funnel(101), *args, chains=32768, seed=20242)
    small = bk.DrGhmcDiag(bk.Funnel(101), *args, chains=96, seed=20242, chain_id0=20000)
    for _ in range(3):
        tb, lb = big.sample()
        ts, ls = small.sample()
        assert torch.equal(tb[20000:20096], ts)
        assert torch.equal(lb[20000:20096], ls)
        assert big.last_stage_lanes[0] == ("P0", 32768)
        assert sum(n * [10, 40, 160][int(t[1])] for t, n in big.last_stage_lanes) == big.last_lane_steps
    assert torch.isfinite(tb).all() and torch.isfinite(lb).all()
    assert torch.equal(big._rho[20000:20096], small._rho)
    np.testing.assert_array_equal(big.rng_state()[:, 20000:20096], small.rng_state())
    # the same draws with the gradient as a SEPARATE counted op per leapfrog step -- the library's own and the user plugin's
    # (bk_target_fn_n) -- 581 launches per draw inside one hipGraph, no host read: theta, rho and stream positions of all
    # 32,768 chains equal the one-launch path bit for bit (round 4)
    for model, kw in ((bk.Funnel(101), dict(path="step")), (funnel_plugin(101), {})):
        o = bk.DrGhmcDiag(model, *args, chains=32768, seed=20242, **kw)
        assert o._dev_counts and not o._one_launch and o._use_graph and o.host_syncs_per_draw == 0
        for _ in range(3):
            to, lo = o.sample()
        assert torch.equal(to, tb) and torch.equal(o._rho, big._rho)
        torch.testing.assert_close(lo, lb, rtol=1e-13, atol=1e-13)   # (kinetic energy summed in another order)
        np.testing.assert_array_equal(o.rng_state(), big.rng_state())
        assert o.last_stage_lanes == big.last_stage_lanes
        del o


@pytest.mark.parametrize("prefetch", [False, True])
def test_hipgraph_replay_equals_eager(ops, prefetch):
    """graph=True replays captured draws; results identical to eager launches.  A captured draw is one
    linear graph: prefetch_rng is accepted with graph=True but the randomness is generated in line
    (a forked capture is slower and makes this ROCm's runtime touch freed memory)."""
    for make in (lambda g, p: bk.HMCDiag(bk.IsoGaussian(128), 0.05, 32, chains=4096, seed=20240, graph=g,
                                         prefetch_rng=p),
                 lambda g, p: bk.HMCDiag(bk.DiagGaussian(np.logspace(0, 1, 40)), 0.05, 7, chains=700, seed=2,
                                         graph=g, prefetch_rng=p, path="step"),
                 lambda g, p: bk.MALA(bk.DiagGaussian(np.logspace(0, 1, 16)), 0.02, chains=512, seed=3, graph=g,
                                      prefetch_rng=p),
                 lambda g, p: bk.MALA(bk.DiagGaussian(np.logspace(0, 1, 48)), 0.02, chains=300, seed=5, graph=g,
                                      prefetch_rng=p),
                 lambda g, p: bk.HMCDiag(bk.TorchModel(lambda Th: -0.5 * (Th * Th).sum(dim=1), 8), 0.2, 6,
                                         chains=256, seed=4, graph=g, prefetch_rng=p)):
        a, b = make(False, False), make(True, prefetch)
        for n in range(9):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), n
            if n in (0, 3, 8):  # the logical stream position is visible between draws
                np.testing.assert_array_equal(a.rng_state(), b.rng_state())
        assert b._graph is not None and not b._prefetch and a._graph is None
        assert a.accept_rate() == b.accept_rate()


def test_rng_prefetch_stream_is_only_a_schedule(ops):
    """Generating draw n+1's randomness on a side stream during draw n changes nothing."""
    lam = np.logspace(0, 2, 64)
    for fused in (False, True):
        a = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 9, chains=3000, seed=8, prefetch_rng=False, path="auto" if fused else "step")
        b = bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 9, chains=3000, seed=8, prefetch_rng=True, path="auto" if fused else "step")
        assert b._prefetch and not a._prefetch
        for n in range(8):
            if n == 4:  # a metric assigned between draws must reach the prefetched kinetic energy
                m = np.linspace(0.9, 1.1, 64)
                a._metric = m
                b._metric = m
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (fused, n)
            np.testing.assert_array_equal(a.rng_state(), b.rng_state())
        assert a.accept_rate() == b.accept_rate()


def test_drghmc_fused_proposal_equals_step_by_step(ops):
    """bk_dr_proposal_funnel (one launch per proposal) vs kick+drift / gradient launches: the
    coordinate sums use the same fixed order, so theta and rho agree bit for bit; energies
    differ only through the kinetic-energy sum order."""
    args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)
    for D, metric in ((11, None), (101, np.linspace(0.9, 1.1, 101)), (129, None)):
        a = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=1500, seed=77, path="step")
        b = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=1500, seed=77, path="auto")
        assert b._fused and not a._fused
        if metric is not None:
            a._metric = metric
            b._metric = metric
        for n in range(6):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert a.last_stage_lanes == b.last_stage_lanes, (D, n)
            assert torch.equal(ta, tb), (D, n)
            np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-12, atol=1e-12)
        assert torch.equal(a._rho, b._rho)
        np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    big = bk.DrGhmcDiag(bk.Funnel(200), *args, chains=64, seed=1)  # falls back to the step path
    big.sample()


def test_checkpoint_resume(ops):
    from tests.sampler_parity import check_checkpoint_resume

    lam = np.logspace(0, 1, 24)
    check_checkpoint_resume(ops, lambda: bk.HMCDiag(bk.DiagGaussian(lam), 0.1, 4, chains=700, seed=2))
    check_checkpoint_resume(ops, lambda: bk.HMCDiag(bk.DiagGaussian(lam), 0.1, 4, chains=700, seed=2,
                                                    path="step", prefetch_rng=False))
    check_checkpoint_resume(ops, lambda: bk.MALA(bk.DiagGaussian(lam), 0.05, chains=700, seed=2))
    check_checkpoint_resume(ops, lambda: bk.DrGhmcDiag(bk.Funnel(21), 3, [0.3, 0.1, 0.03], [3, 9, 27], 0.3,
                                                       chains=700, seed=2))


def test_dense_metric_hmc_vs_oracle(ops):
    from tests.sampler_parity import check_dense_metric_hmc

    check_dense_metric_hmc(ops, C=300, D=48)
    check_dense_metric_hmc(ops, C=130, D=130, draws=3)


def test_tempered_smc_binomial_moments(ops):
    """test/test_tempered_smc.py:8-30 with many more particles on the per-slot Philox streams (the pathwise pin to the
    reference's own stream is test_tempered_smc_reference_stream_vs_reference_golden)."""
    from tests.sampler_parity import check_smc_binomial

    a = check_smc_binomial(ops, 8192, 15, bk.metropolis_kernel(0.5), mean_atol=0.006, var_atol=0.0012)
    b = check_smc_binomial(ops, 8192, 15, bk.metropolis_kernel(0.5), mean_atol=0.006, var_atol=0.0012)
    assert torch.equal(a.thetas, b.thetas)  # reproducible: every slot owns a Philox stream
    check_smc_binomial(ops, 8192, 10, bk.mala_kernel(0.2, 2), mean_atol=0.006, var_atol=0.0012)


@pytest.mark.parametrize("path", ["one_launch", "counted_steps"])
@pytest.mark.parametrize("name", ["drghmc_funnel11_k3", "drghmc_funnel101_cfg4", "drghmc_funnel17_k4",
                                  "drghmc_funnel33_k2_metric_noretry", "drghmc_funnel129_k3", "drghmc_funnel130_k2"])
def test_funnel_fixtures_bit_identical_to_the_canonical_order_oracle(ops, name, path):
    """SURVEY 8c's funnel bar is theta rel 1e-9 over <= 50 draws.  HIP against the oracle that sums in the library's
    canonical order and uses the library's exp (the same rounded operations on both sides): theta and momentum
    BIT-IDENTICAL over all draws of every funnel fixture, decisions and stream state exact -- through the one-launch
    proposal kernel and through counted leapfrog steps."""
    from tests.sampler_parity import check_funnel_vs_canonical_oracle

    extra = {} if path == "one_launch" else dict(path="step")
    check_funnel_vs_canonical_oracle(name, ops, **extra)


@pytest.mark.parametrize("source", ["np.random", "RandomState", "replay"])
@pytest.mark.parametrize("name", ["smc_ref_binomial", "smc_gauss5_m512", "smc_gauss3_m2048"])
def test_tempered_smc_reference_stream_vs_reference_golden(ops, name, source):
    """bayes_kit/smc.py:12-89 run by the REAL reference under np.random.seed(s) (tests/golden/make_golden.py::
    run_smc_case) against the device SMC fed the same stream: moved particles, ancestor indices (choice's cdf in
    np.cumsum's order, np.sum's pairwise total) and resampled particles BIT-EXACT after every temperature."""
    from tests.sampler_parity import check_smc_reference_stream

    check_smc_reference_stream(ops, name, source)


def test_tempered_smc_reference_stream_vs_oracle_other_seeds(ops):
    """The same comparison against oracle/smc.py (pinned to the fixtures on the CPU) on seeds and sizes no fixture
    holds, D even and odd."""
    import torch

    from oracle import models as om
    from oracle import smc as osmc

    for seed, M, D, N, scale in [(5, 300, 2, 5, 0.4), (6, 1000, 7, 3, 0.2), (7, 64, 1, 9, 0.8)]:
        g = np.random.default_rng(seed)
        y, prec = g.normal(size=D), np.logspace(0, 1, D)
        host = om.GaussPriorLik(y, prec, prior_scale=1.3)
        init = g.normal(size=(M, D)) * 1.3
        stream = osmc.NumpyLegacySource(np.random.RandomState(seed))
        o = osmc.TemperedLikelihoodSMC(host, M, N, lambda i: init[i], osmc.metropolis_kernel(scale, stream), stream)
        yt, pt = (torch.as_tensor(v, dtype=torch.float64, device=ops.device) for v in (y, prec))
        model = bk.TorchPriorLikelihoodModel(lambda T: host._c0 * (T * T).sum(dim=1),
                                             lambda T: -0.5 * (pt * ((T - yt) * (T - yt))).sum(dim=1), D)
        s = bk.TemperedLikelihoodSMC(model, M, N, init, bk.metropolis_kernel(scale), seed=np.random.RandomState(seed), ops=ops)
        for n in range(1, N + 1):
            o.transition(n)
            s.transition(n)
            assert np.array_equal(np.asarray(s._idx.cpu()), o.idxs), (seed, n)
            assert np.array_equal(np.asarray(s.thetas.cpu()), o.thetas), (seed, n)


def test_logistic_regression_target_and_annealed_smc(ops):
    """Config-5 pieces (no reference oracle): MFMA logistic target vs NumPy, HMC on it vs the
    oracle, then likelihood-annealed SMC with a Langevin move recovers the coefficients."""
    from tests.sampler_parity import check_logistic_target

    check_logistic_target(ops)
    model, tstar = check_logistic_target(ops, N=4000, D=40, C=257)
    M = 2048
    init = np.random.default_rng(0).normal(size=(M, 40)) * 2.0  # draws from the N(0, 2^2) prior
    smc = bk.TemperedLikelihoodSMC(model, M, 12, init, bk.mala_kernel(0.004, 3), seed=3)
    smc.run()
    post = smc.thetas.mean(dim=0).cpu().numpy()
    assert np.corrcoef(post, tstar)[0, 1] > 0.9
    # the same with HMC moves under a dense metric (MFMA GEMMs in the move kernel as well)
    Md = np.eye(40) * 0.04
    smc2 = bk.TemperedLikelihoodSMC(model, M, 12, init, bk.hmc_kernel(0.3, 5, metric_dense=Md), seed=3)
    smc2.run()
    post2 = smc2.thetas.mean(dim=0).cpu().numpy()
    assert np.corrcoef(post2, tstar)[0, 1] > 0.9
    assert np.corrcoef(post2, post)[0, 1] > 0.9  # two move kernels, same posterior (Monte Carlo noise apart)


@pytest.mark.parametrize("R,K,C", [(128, 16, 128), (300, 70, 200), (1000, 512, 256), (40, 3000, 130), (128, 8192, 128)])
def test_rectangular_mfma_gemm(ops, R, K, C):
    rng = np.random.default_rng(R + K + C)
    A, X = rng.normal(size=(R, K)), rng.normal(size=(K, C))
    Y = torch.empty((R, C), dtype=torch.float64, device=ops.device)
    ops.gemm_chains(torch.from_numpy(A).to(ops.device), torch.from_numpy(X).to(ops.device), Y)
    np.testing.assert_allclose(Y.cpu().numpy(), A @ X, rtol=1e-12, atol=1e-12 * np.sqrt(K))
    # split-K with a caller-owned workspace gives the same product
    work = ops.gemm_chains_work(R, K, C)
    assert (work is not None) == (K >= 2048)  # (the split depends on R and K only: long inner dimensions, few row blocks)
    Y2 = torch.empty_like(Y)
    ops.gemm_chains(torch.from_numpy(A).to(ops.device), torch.from_numpy(X).to(ops.device), Y2, work)
    np.testing.assert_allclose(Y2.cpu().numpy(), A @ X, rtol=1e-12, atol=1e-12 * np.sqrt(K))
    if work is not None:
        # ... and a too-small workspace is refused rather than answered with another split
        with pytest.raises(bk._lib.BkHipError):
            ops.gemm_chains(torch.from_numpy(A).to(ops.device), torch.from_numpy(X).to(ops.device), Y2, work[: work.numel() // 2])


def test_split_gemm_does_not_depend_on_how_many_chains_share_the_call(ops):
    """SURVEY 8e for the logistic target's X^T r: a chain's column of the split-K GEMM is the same bits whether it is
    evaluated among 3,000 chains (two column blocks), among 256 or alone in an odd-sized shard."""
    R, K, C = 96, 40_000, 3000
    g = torch.Generator(device=ops.device)
    g.manual_seed(1)
    A = torch.randn((R, K), dtype=torch.float64, device=ops.device, generator=g)
    X = torch.randn((K, C), dtype=torch.float64, device=ops.device, generator=g)
    Y = torch.empty((R, C), dtype=torch.float64, device=ops.device)
    ops.gemm_chains(A, X, Y, ops.gemm_chains_work(R, K, C))
    for c0, cw in ((0, 256), (512, 1024), (2047, 131), (2999, 1), (1024, 1976)):
        Xs = X[:, c0:c0 + cw].contiguous()
        Ys = torch.empty((R, cw), dtype=torch.float64, device=ops.device)
        ops.gemm_chains(A, Xs, Ys, ops.gemm_chains_work(R, K, cw))
        assert torch.equal(Ys, Y[:, c0:c0 + cw]), (c0, cw)


def test_mala_rng_prefetch_is_only_a_schedule(ops):
    lam = np.logspace(0, 2, 48)
    a = bk.MALA(bk.DiagGaussian(lam), 0.004, chains=2500, seed=8, prefetch_rng=False)
    b = bk.MALA(bk.DiagGaussian(lam), 0.004, chains=2500, seed=8, prefetch_rng=True)
    assert b._prefetch and not a._prefetch
    for _ in range(8):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb)
        np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    assert 0.2 < a.accept_rate() == b.accept_rate()


@pytest.mark.parametrize("C,D", [(1, 1), (3, 2), (65, 7), (130, 129), (257, 33), (64, 200)])
def test_odd_shapes_all_samplers_vs_oracle(ops, C, D):
    """Ragged shapes (odd chain counts, D not a multiple of any tile, D = 1): every sampler on
    the device vs the oracle chain by chain, built-in fast paths on and off."""
    from oracle import models as om
    from oracle import samplers as osamp

    lam = np.logspace(0, 1, D)
    seed = 1000 + C + D
    chains_to_check = sorted({0, C // 2, C - 1})

    def compare(dev_sampler, make_oracle, draws, exact=True, tol=None):
        outs = [dev_sampler.sample() for _ in range(draws)]
        for c in chains_to_check:
            o = make_oracle(c)
            for n, (th, lp) in enumerate(outs):
                oth, olp = o.sample()
                t = tol(n) if tol is not None else dict(rtol=1e-7, atol=1e-9)
                if exact:
                    assert np.array_equal(th[c].cpu().numpy(), oth), (type(dev_sampler).__name__, C, D, c)
                else:
                    np.testing.assert_allclose(th[c].cpu().numpy(), oth, **t)
                np.testing.assert_allclose(lp[c].item(), olp, **(dict(rtol=1e-11, atol=1e-12) if exact else t))

    key = lambda c: np.random.Philox(key=[seed, c])
    for fused in (True, False):
        compare(bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=C, seed=seed, path="auto" if fused else "step"),
                lambda c: osamp.HMCDiag(om.DiagGaussian(lam), 0.05, 5, seed=key(c)), 4)
    compare(bk.MALA(bk.DiagGaussian(lam), 0.01, chains=C, seed=seed),
            lambda c: osamp.MALA(om.DiagGaussian(lam), 0.01, seed=key(c)), 4)
    args = (3, [0.3, 0.1, 0.03], [2, 5, 11], 0.4)
    compare(bk.DrGhmcDiag(bk.DiagGaussian(lam), *args, chains=C, seed=seed),
            lambda c: osamp.DrGhmcDiag(om.DiagGaussian(lam), *args, seed=key(c)), 6)
    if D >= 2:
        for fused in (True, False):
            compare(bk.DrGhmcDiag(bk.Funnel(D), *args, chains=C, seed=seed, path="auto" if fused else "step"),
                    lambda c: osamp.DrGhmcDiag(om.Funnel(D), *args, seed=key(c)), 5, exact=False, tol=funnel_tol)
    # dense metric on a ragged shape (MFMA tiles with bounds checks)
    if D <= 64:
        M = np.eye(D) + 0.1 * np.outer(np.ones(D), np.ones(D)) / D
        compare(bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 4, chains=C, seed=seed, metric_dense=M),
                lambda c: osamp.HMCDense(om.DiagGaussian(lam), 0.05, 4, M, seed=key(c)), 3, exact=False)


def test_moments_many_chains_hmc_and_mala(ops):
    """Statistical checks in the spirit of test/test_hmc.py:38-51 and test/test_mala.py:9-41,
    with thousands of chains at once: draws match the target's mean and variance."""
    lam = np.array([1.0, 4.0, 0.25])
    for s, burn, keep in (
        (bk.HMCDiag(bk.DiagGaussian(lam), 0.25, 10, chains=8192, seed=12), 20, 30),
        (bk.HMCDiag(bk.DiagGaussian(lam), 0.25, 10, chains=8192, seed=12, path="step"), 20, 30),
        (bk.MALA(bk.DiagGaussian(lam), 0.2, chains=8192, seed=13), 150, 50),
    ):
        acc = []
        for n in range(burn + keep):
            th, _ = s.sample()
            if n >= burn:
                acc.append(th.clone())
        x = torch.stack(acc).reshape(-1, 3).cpu().numpy()
        np.testing.assert_allclose(x.mean(axis=0), 0.0, atol=0.02)
        np.testing.assert_allclose(x.var(axis=0), 1.0 / lam, rtol=0.04)
        assert 0.3 < s.accept_rate() <= 1.0


def test_graph_replay_is_automatic_for_small_builtin_problems(ops):
    small = bk.HMCDiag(bk.IsoGaussian(128), 0.05, 32, chains=4096, seed=1)
    big = bk.HMCDiag(bk.IsoGaussian(1024), 0.05, 2, chains=8192, seed=1)
    torchm = bk.HMCDiag(bk.TorchModel(lambda Th: -0.5 * (Th * Th).sum(dim=1), 8), 0.2, 3, chains=64, seed=1)
    assert small._use_graph and not small._prefetch
    assert not big._use_graph and big._prefetch
    assert not torchm._use_graph
    for _ in range(2):
        small.sample()
    assert small._graph is not None  # captured right after the single eager warm-up draw


def test_metric_assignment_survives_graph_replay(ops):
    """_metric assigned after capture must reach the replayed kernels (in-place update, or a
    re-capture when there was no metric before)."""
    lam = np.logspace(0, 1, 16)
    m = np.linspace(0.9, 1.1, 16)
    for first_metric in (None, np.ones(16)):
        a = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=512, seed=3, graph=True, metric_diag=first_metric)
        b = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=512, seed=3, graph=False, prefetch_rng=False,
                       metric_diag=first_metric)
        for n in range(7):
            if n == 4:
                a._metric = m
                b._metric = m
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (first_metric is None, n)


def test_metropolis_accept_functions_and_moments(ops):
    """metropolis.py:12-76 on device streams + a many-chain random-walk Metropolis run
    (moment test in the spirit of test_metropolis.py:106-168)."""
    from bayes_kit_amd.metropolis import ChainRng, metropolis_accept_test, metropolis_hastings_accept_test

    C = 300
    rng = ChainRng(31, C, ops=ops)
    gens = [np.random.Generator(np.random.Philox(key=[31, c])) for c in range(C)]
    logu = np.array([np.log(g.uniform()) for g in gens])
    # a few ulp either side of the boundary (the device log is within 1 ulp of the host's, as
    # different hosts' logs are of each other: DESIGN.md parity table)
    k = np.arange(C) % 2
    delta = np.where(k == 0, logu * (1 - 2e-15), logu * (1 + 2e-15))
    zero = torch.zeros(C, dtype=torch.float64, device=ops.device)
    got = metropolis_accept_test(torch.from_numpy(delta).to(ops.device), zero, rng)
    assert got.cpu().tolist() == [int(i) == 0 for i in k]
    logu2 = np.array([np.log(g.uniform()) for g in gens])
    fwd, rev, lp_p = np.linspace(-1, 1, C), np.linspace(0.5, -2, C), np.linspace(-3, 0.2, C)
    got = metropolis_hastings_accept_test(lp_p, zero, fwd, rev, rng)
    assert got.cpu().tolist() == [bool(l < (p - 0.0) + (r - f)) for l, p, f, r in zip(logu2, lp_p, fwd, rev)]

    C, D = 4096, 3
    prop = ChainRng(77, C, ops=ops)
    s = bk.Metropolis(bk.IsoGaussian(D), lambda Th: prop.normal(Th, 1.2), chains=C, seed=5)
    for _ in range(100):
        s.sample()
    draws = torch.stack([s.sample()[0] for _ in range(50)])
    assert 0.2 < s.accept_rate() < 0.6, s.accept_rate()
    # 4096 independent chains x 50 (autocorrelated) draws x 3 dims: s.e. ~0.01 on both moments
    assert abs(float(draws.mean())) < 0.05 and abs(float(draws.var()) - 1.0) < 0.06, (draws.mean(), draws.var())


def test_user_plugin_target_through_the_c_abi(ops):
    """examples/plugin_target: a user-compiled target loaded with bk.CTarget (plugin ABI
    bk_target_fn).  Its outputs and an HMC run on it are checked against the NumPy statement
    of the same density (tests/host_models.Ar1) driven by the oracle sampler."""
    import ctypes
    import os

    from oracle import samplers as osamp
    from tests.host_models import Ar1

    class Params(ctypes.Structure):
        _fields_ = [("a", ctypes.c_double), ("s2", ctypes.c_double)]

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    C, D, a, s2 = 70, 16, 0.6, 0.8
    tgt = bk.CTarget(os.path.join(root, "examples", "plugin_target", "libar1_target.so"), "ar1_target", D,
                     Params(a, s2))
    host = Ar1(D, a, s2)
    Th = torch.randn((D, C), dtype=torch.float64, device=ops.device).t()  # (C, D) view, strides (1, C)
    lp, g = tgt.log_density_gradient(Th)
    for c in (0, 1, C - 1):
        hlp, hg = host.log_density_gradient(Th[c].cpu().numpy())
        assert np.array_equal(g[c].cpu().numpy(), hg) and float(lp[c]) == hlp
    assert torch.equal(tgt.log_density(Th), lp)
    s = bk.HMCDiag(tgt, 0.1, 6, chains=C, seed=909)
    th0 = s._theta.cpu().numpy()
    draws = [tuple(x.cpu().numpy() for x in s.sample()) for _ in range(12)]
    assert 0.3 < s.accept_rate() <= 1.0
    for c in (0, 33, C - 1):
        o = osamp.HMCDiag(host, 0.1, 6, seed=np.random.Philox(key=[909, c]))
        assert np.array_equal(o._theta, th0[c])
        for n in range(12):
            oth, olp = o.sample()
            assert np.array_equal(oth, draws[n][0][c]), (c, n)
            np.testing.assert_allclose(olp, draws[n][1][c], rtol=1e-12, atol=1e-12)
    with pytest.raises(bk._lib.BkHipError):
        bk.CTarget(os.path.join(root, "examples", "plugin_target", "libar1_target.so"), "no_such_symbol", D)


def test_reference_test_behaviours_on_device(ops):
    """test/test_iat.py:72-80, test_metropolis.py:19-103, test_theta_initialization.py:17-54,
    test_tempered_smc.py:8-30 in this repo's words, through the HIP library."""
    from tests import dropin_behaviours as db

    db.check_end_pos_pairs(ops)
    db.check_accept_tests_with_host_rng(ops)
    db.check_theta_initialization(ops)
    db.check_smc_with_reference_style_model(ops)


@pytest.mark.parametrize("alg", ["hmc", "mala", "drghmc"])
def test_long_run_stays_bit_identical_to_the_oracle(ops, alg):
    """4000 draws x 40 dims per chain = 160,000+ normals per stream (tens of ziggurat tail and
    thousands of wedge draws per chain, 40,000 Philox buffer wraps, the wavefront-per-chain
    generator resuming at every buffer position): theta bit-identical to the NumPy oracle at
    every draw, final stream state equal."""
    from oracle import models as om
    from oracle import samplers as osamp

    C, D, N, seed = 96, 40, 4000, 8675309
    lam = np.logspace(0, 0.7, D)
    if alg == "hmc":
        s = bk.HMCDiag(bk.DiagGaussian(lam), 0.2, 3, chains=C, seed=seed, path="step")
        mk = lambda sd: osamp.HMCDiag(om.DiagGaussian(lam), 0.2, 3, seed=sd)  # noqa: E731
    elif alg == "mala":
        s = bk.MALA(bk.DiagGaussian(lam), 0.05, chains=C, seed=seed)
        mk = lambda sd: osamp.MALA(om.DiagGaussian(lam), 0.05, seed=sd)  # noqa: E731
    else:
        s = bk.DrGhmcDiag(bk.DiagGaussian(lam), 2, [0.5, 0.2], [2, 4], 0.3, chains=C, seed=seed)
        mk = lambda sd: osamp.DrGhmcDiag(om.DiagGaussian(lam), 2, [0.5, 0.2], [2, 4], 0.3, seed=sd)  # noqa: E731
    watch = [0, 41, C - 1]
    got = torch.empty((N, len(watch), D), dtype=torch.float64, device=ops.device)
    for n in range(N):
        th, _ = s.sample()
        got[n] = th[watch]
    got = got.cpu().numpy()
    state = s.rng_state()
    for j, c in enumerate(watch):
        o = mk(np.random.Philox(key=[seed, c]))
        for n in range(N):
            oth, _ = o.sample()
            assert np.array_equal(oth, got[n, j]), (alg, c, n)
        st = o._rng.bit_generator.state
        assert [int(v) for v in st["state"]["counter"]] == [int(v) for v in state[2:6, c]]
        assert int(st["buffer_pos"]) == int(state[10, c])


def test_randomised_sampler_configurations_against_the_oracle(ops):
    """A fixed-seed slice of tests/soak_samplers.py: random algorithm, target, model provider (library target,
    autograd, user gradient layouts, compiled plugin), dims (both generator kernels), odd / even chain counts, metric, fused / step-by-step, hipGraph and RNG-prefetch
    switches; watched chains bit-identical to the oracle at every draw, stream states equal."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak_samplers.py")
    spec = importlib.util.spec_from_file_location("soak_samplers", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(424242)
    seen = set()
    for it in range(120):
        seen.add(str(mod.one(rng, it)))
    assert seen == {"hmc", "mala", "drghmc", "metropolis", "drfunnel", "hmcfunnel"}
    assert set(mod.PROVIDERS_SEEN) == {"builtin", "torch", "row", "strided", "plugin"}, mod.PROVIDERS_SEEN


def test_plain_c_host_program_equals_the_python_driver(ops):
    """examples/c_host/hmc_main.c drives a whole many-chain HMC run through the C ABI alone (gcc, no
    Python, no torch).  Its final state must be bit-identical to bayes_kit_amd.HMCDiag issuing the same
    calls through ctypes, and so (via the other tests) to the reference."""
    import os
    import subprocess

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    C, D, L, draws, seed = 130, 40, 5, 6, 77
    out = subprocess.run([os.path.join(root, "examples", "c_host", "hmc_main"), str(C), str(D), str(L), str(draws),
                          str(seed)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    accept = float(lines[0].split()[-1])
    cols = {int(ln.split("]")[0].split("[")[1]): np.array([int(w, 16) for w in ln.split()[1:]], dtype=np.uint64)
            for ln in lines[1:]}
    s = bk.HMCDiag(bk.DiagGaussian(np.linspace(1.0, 2.0, D)), 0.05, L, chains=C, seed=seed, path="step",
                   prefetch_rng=False, graph=False)
    for _ in range(draws):
        th, _ = s.sample()
    th = th.cpu().numpy()
    for c, words in cols.items():
        assert np.array_equal(th[c].view(np.uint64), words), c
    assert abs(accept - s.accept_rate()) < 1e-6


def test_placement_tuning_is_only_a_choice_of_buffers(ops):
    """tune_placement picks which scratch allocation plays which role by timing; draws are unchanged."""
    lam = np.logspace(0, 1, 40)
    a = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 7, chains=700, seed=2, path="step", graph=False,
                   tune_placement=False)
    b = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 7, chains=700, seed=2, path="step", graph=False,
                   tune_placement=True)
    assert a.placement is None and b.placement["assignments_tried"] == bk.HMCDiag.TUNE_PLACEMENT_TRIALS
    for n in range(6):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb), n
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())


def test_placement_tuning_leaves_the_users_allocator_alone(ops):
    """the spare candidates of the placement tuning come from the driver (hipMalloc) and go back
    to it; PyTorch's caching allocator is not flushed (a block the USER freed stays cached), nothing of the sampler's
    own scratch is leaked, and the chosen arrays survive the spares."""
    import gc

    lam = np.logspace(0, 1, 64)
    mk = lambda: bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 3, chains=4096, seed=2, path="step", graph=False,  # noqa: E731
                            tune_placement=True)
    mk().sample()   # (first use: code objects, the runtime's own pools)
    gc.collect()
    torch.cuda.synchronize()
    user = torch.empty(64 << 20, dtype=torch.uint8, device=ops.device)   # a user's block ...
    del user                                                             # ... freed: now cached by PyTorch
    reserved = torch.cuda.memory_reserved()
    raw = bk._lib.RawDeviceArray
    gc.collect()
    base = raw.live          # (samplers of earlier tests that are still referenced somewhere keep theirs)
    s = mk()
    assert s.placement["assignments_tried"] == bk.HMCDiag.TUNE_PLACEMENT_TRIALS
    assert torch.cuda.memory_reserved() >= reserved     # (empty_cache() would have returned the user's block)
    assert 0 <= raw.live - base <= bk.HMCDiag.TUNE_PLACEMENT_SPARES  # only the spares that were CHOSEN are still allocated
    t0, _ = s.sample()
    t1, _ = s.sample()
    assert torch.isfinite(t0).all() and torch.isfinite(t1).all()
    del s, t0, t1
    gc.collect()
    assert raw.live == base    # ... and they go back to the driver with the sampler


def test_mala_placement_tuning_is_only_a_choice_of_buffers(ops):
    lam = np.logspace(0, 1, 48)
    a = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=600, seed=5, graph=False, tune_placement=False, two_pass=False)
    b = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=600, seed=5, graph=False, tune_placement=True, two_pass=False)
    assert a.placement is None and b.placement["assignments_tried"] == bk.MALA.TUNE_PLACEMENT_TRIALS
    for n in range(6):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb), n
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())


@pytest.mark.parametrize("C,D", [(2, 32), (34, 33), (130, 129), (64, 200), (600, 513), (48, 1024), (18, 1000), (4096, 128)])
def test_mala_two_pass_equals_step_by_step(ops, C, D):
    """bk_mala_step (sums, decision, select and the next proposal in one kernel, state array rebound
    every draw) against the separate kernels: same draws, same accept masks, same logical stream
    position after every draw; both RNG schedules."""
    lam = np.logspace(0, 1.5, D)
    eps = 0.3 / D
    for prefetch in (False, True):
        a = bk.MALA(bk.DiagGaussian(lam), eps, chains=C, seed=77, two_pass=False, prefetch_rng=False, graph=False)
        b = bk.MALA(bk.DiagGaussian(lam), eps, chains=C, seed=77, two_pass=True, prefetch_rng=prefetch, graph=False)
        assert b.path.startswith("two-pass") and a.path == "step-by-step"
        kept = []
        for n in range(7):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (prefetch, n)
            assert torch.equal(a.last_accept, b.last_accept)
            kept.append((tb, tb.clone()))
            if n in (0, 2, 6):
                np.testing.assert_array_equal(a.rng_state(), b.rng_state())
        assert 0.05 < a.accept_rate() == b.accept_rate()
        for t, c in kept:  # returned draws are never written again (the state array is rebound)
            assert torch.equal(t, c)
        assert torch.equal(a._log_p_grad_theta, b._log_p_grad_theta) and torch.equal(a._log_p_theta, b._log_p_theta)


@pytest.mark.parametrize("C,D", [(2, 32), (130, 129), (64, 200), (48, 1024), (18, 1000), (4096, 128)])
def test_mala_step_kernel_with_the_density_inlined_equals_the_model_opaque_pair(ops, C, D):
    """A separable density the library can inline (built-in Gaussians, an elementwise source, a traced PyTorch function): the
    step kernel recomputes both gradients and stores none (model.bk_mala_step; 56 D bytes per chain-draw) -- the same draws,
    masks, stream positions and, on request, cached gradient as the model-opaque pair {gradient op, bk_mala_step}."""
    lam = np.logspace(0, 1.5, D)
    lam_d = torch.from_numpy(lam).to(ops.device)
    eps = 0.3 / D
    src = "__device__ __forceinline__ void bk_term(double th, i64 d, const double* lam, double& term, double& grad) {\n" \
          "  const double lt = lam[d] * th; term = -0.5 * (th * lt); grad = -lt; }\n"
    models = [lambda: bk.DiagGaussian(lam), lambda: bk.IsoGaussian(D), lambda: bk.CTarget.from_source(src, D, params=lam_d)]
    if D <= 200:
        models.append(lambda: bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam_d).sum(dim=1), D, compile=True))
    for mi, model_of in enumerate(models):
        for kw in (dict(prefetch_rng=False, graph=False), dict(prefetch_rng=True, graph=False), dict(graph=True)):
            a = bk.MALA(model_of(), eps, chains=C, seed=78, two_pass=True, path="step", prefetch_rng=False, graph=False)
            b = bk.MALA(model_of(), eps, chains=C, seed=78, two_pass=True, **kw)
            assert b._sep_step and not a._sep_step and "recomputed" in b.path, (mi, b.path)
            for n in range(6):
                ta, la = a.sample()
                tb, lb = b.sample()
                assert torch.equal(ta, tb) and torch.equal(la, lb), (mi, kw, n)
                assert torch.equal(a.last_accept, b.last_accept)
                if n == 2:
                    assert torch.equal(a._log_p_grad_theta, b._log_p_grad_theta)
                    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
            assert 0.05 < a.accept_rate() == b.accept_rate()
            assert torch.equal(a._log_p_grad_theta, b._log_p_grad_theta) and torch.equal(a._log_p_theta, b._log_p_theta)
            if mi == 0 and not kw.get("graph"):   # checkpoint of the inlined path -> resumed on the model-opaque one
                sd = b.state_dict()
                c = bk.MALA(model_of(), eps, chains=C, seed=1, two_pass=True, path="step", graph=False)
                c.load_state_dict(sd)
                for n in range(3):
                    tb, lb = b.sample()
                    tc, lc = c.sample()
                    assert torch.equal(tb, tc) and torch.equal(lb, lc), n


def test_mala_inlined_step_narrow_workgroups_give_the_same_draws():
    """BK_MALA_STEP_PAIRS=4 (8 chains per workgroup, one wavefront per SIMD: the shape measured beside the generator in
    profiles/r5_mala.md) keeps the summation order: the same draws as the model-opaque pair.  Own process: the choice is read once."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np, torch, sys\n"
        "sys.path.insert(0, 'bayes-kit_amd')\n"
        "import bayes_kit_amd as bk\n"
        "for C, D in ((130, 129), (48, 1024), (4096, 100)):\n"
        "    lam = np.logspace(0, 1.5, D)\n"
        "    a = bk.MALA(bk.DiagGaussian(lam), 0.3 / D, chains=C, seed=5, two_pass=True, path='step')\n"
        "    b = bk.MALA(bk.DiagGaussian(lam), 0.3 / D, chains=C, seed=5, two_pass=True)\n"
        "    assert b._sep_step and not a._sep_step\n"
        "    for n in range(6):\n"
        "        ta, la = a.sample(); tb, lb = b.sample()\n"
        "        assert torch.equal(ta, tb) and torch.equal(la, lb), (C, D, n)\n"
        "    assert np.array_equal(a.rng_state(), b.rng_state())\n"
        "print('narrow ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, BK_MALA_STEP_PAIRS="4"), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "narrow ok" in out.stdout, out.stderr[-3000:]


def test_whole_draw_hmc_rebinds_its_state_and_keeps_returned_draws(ops):
    """The one-pass HMC draw writes the blend of state and proposal to a fresh array that becomes the
    state (bk_blend_columns): draws handed out earlier are never written again, not by later draws and
    not by load_state_dict(); the chain equals the in-place path's bit for bit."""
    lam = np.logspace(0, 1, 48)
    a = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=130, seed=9, graph=False, path="step")
    b = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=130, seed=9, graph=False, path="auto")
    assert b._fused_draw and not a._fused
    kept, sd = [], None
    for n in range(8):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb), n
        kept.append((tb, tb.clone()))
        if n == 3:
            sd = b.state_dict()
    tail = [t.clone() for t, _ in kept[4:]]
    b.load_state_dict(sd)  # back to the state after draw 3: draws 4.. are produced again
    for n in range(4, 8):
        tb, _ = b.sample()
        assert torch.equal(tb, tail[n - 4]), n
    for t, c in kept:
        assert torch.equal(t, c)
    assert 0.1 < a.accept_rate() < 1.0


@pytest.mark.parametrize("graph", [False, True])
def test_identity_metric_is_not_multiplied_in_but_changes_nothing(ops, graph):
    """A metric of ones (the reference's default, hmc.py:22) is an exact identity: the one-pass draw runs
    without it; a metric assigned later (also under hipGraph replay, which must be captured again) is
    multiplied in.  Every variant equals the step-by-step path, which always multiplies."""
    D, C = 40, 130
    lam = np.logspace(0, 1, D)
    mk = lambda fused, metric: bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 4, metric_diag=metric, chains=C, seed=21,
                                          graph=graph and fused, path="auto" if fused else "step")
    a, b, c = mk(False, np.ones(D)), mk(True, np.ones(D)), mk(True, None)
    assert b._fused_draw and b._metric_identity and a._metric_dev is not None
    for n in range(10):
        if n == 5:
            for s in (a, b, c):
                s._metric = np.linspace(0.8, 1.3, D)
            assert not b._metric_identity
        ta, la = a.sample()
        for s in (b, c):
            t, l = s.sample()
            assert torch.equal(ta, t) and torch.equal(la, l), n


def test_mala_step_kernel_against_its_cpu_statement(ops):
    """The kernel alone on random inputs (ragged last block, odd D, in-place theta_out, no next
    proposal) against tests/fake_ops.py's NumPy statement of bk_mala_step."""
    from tests.fake_ops import FakeOps

    fake = FakeOps()
    rng = np.random.default_rng(5)
    for C, D, inplace, with_z in [(34, 33, False, True), (16, 64, True, True), (50, 1000, False, False),
                                  (300, 257, True, True), (2, 1024, False, True)]:
        eps = 0.01
        s2 = float(np.sqrt(2 * eps))
        th = rng.normal(size=(D, C))
        g = -th * rng.uniform(0.5, 2.0, size=(D, 1))
        thp = th + eps * g + s2 * rng.normal(size=(D, C))
        gp = -thp * rng.uniform(0.5, 2.0, size=(D, 1))
        lp, lpp, logu = rng.normal(size=C), rng.normal(size=C), np.log(rng.uniform(size=C))
        dp = (D + 7) // 8 * 8
        zt = rng.normal(size=(C, dp))

        def run(o, dev):
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
            a = dict(th=t(th), g=t(g), thp=t(thp), gp=t(gp), lp=t(lp), lpp=t(lpp), logu=t(logu), zt=t(zt))
            out = a["th"] if inplace else torch.full_like(a["th"], float("nan"))
            mask = torch.zeros(C, dtype=torch.uint8, device=dev)
            ret = torch.zeros(C, dtype=torch.float64, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            o.mala_step(a["th"], out, a["g"], a["thp"], a["gp"], a["lp"], a["lpp"], a["logu"],
                        a["zt"] if with_z else None, eps, s2, mask, ret, cnt)
            return [x.cpu().numpy() for x in (out, a["g"], a["thp"], a["lp"], ret, mask, cnt)]

        got, want = run(ops, ops.device), run(fake, "cpu")
        for k, (x, y) in enumerate(zip(got, want)):
            assert np.array_equal(x, y), (C, D, inplace, with_z, k)
        assert 0 < int(got[6][0]) < C or C <= 2


def test_mala_two_pass_checkpoint_cache_refresh_and_graph(ops):
    lam = np.logspace(0, 1, 64)
    make = lambda **kw: bk.MALA(bk.DiagGaussian(lam), 0.01, chains=700, seed=2, two_pass=True, graph=False, **kw)  # noqa: E731
    check_checkpoint_resume(ops, make)
    # load_state_dict right after sample(), no synchronisation in between: the generator queued on the
    # side stream for the next draw must not overwrite the restored stream table
    a = make()
    for _ in range(3):
        a.sample()
    sd = a.state_dict()
    want = [a.sample() for _ in range(4)]
    for _ in range(5):
        a.sample()
        a.load_state_dict(sd)
        for tw, lw in want:
            tg, lg = a.sample()
            assert torch.equal(tw, tg) and torch.equal(lw, lg)
        a.sample()
    # a state edited from outside + refresh_cache() == a sampler constructed at that state
    b = make()
    for _ in range(3):
        b.sample()
    th = b._theta.clone() * 0.5
    b._theta_dc.copy_(th.t())
    b.refresh_cache()
    st = b.rng_state().copy()
    c = make()
    c._theta_dc.copy_(th.t())
    c._rng_state.copy_(torch.from_numpy(st.view(np.int64)).to(ops.device))
    c.refresh_cache()
    for n in range(4):
        tb, lb = b.sample()
        tc, lc = c.sample()
        assert torch.equal(tb, tc) and torch.equal(lb, lc), n
    # one serial hipGraph per draw replays the two-pass draw (state updated in place)
    e = bk.MALA(bk.DiagGaussian(lam), 0.01, chains=700, seed=2, two_pass=True, graph=False, prefetch_rng=False)
    g = bk.MALA(bk.DiagGaussian(lam), 0.01, chains=700, seed=2, two_pass=True, graph=True)
    for n in range(8):
        te, le = e.sample()
        tg, lg = g.sample()
        assert torch.equal(te, tg) and torch.equal(le, lg), n
    assert g._graph is not None
    np.testing.assert_array_equal(e.rng_state(), g.rng_state())


@pytest.mark.parametrize("C,D", [(2, 1), (65, 7), (64, 32), (130, 129), (257, 33), (96, 1000), (512, 1024), (4096, 128)])
def test_hmc_whole_draw_kernel_equals_step_by_step(ops, C, D):
    """bk_hmc_draw_gaussian (trajectory + kin0 + kin1 + end-point log density in one pass, momentum
    consumed chain-major from the generator) against the separate kernels: theta AND the returned
    joint log density bit-identical, same accept masks and stream positions; with and without a
    metric, diag and iso targets, both RNG schedules."""
    lam = np.logspace(0, 1.2, D)
    for model_of, metric in ((lambda: bk.DiagGaussian(lam), np.linspace(0.9, 1.1, D)), (lambda: bk.DiagGaussian(lam), None),
                             (lambda: bk.IsoGaussian(D), None)):
        for prefetch in (False, True):
            kw = dict(metric_diag=metric, chains=C, seed=31, graph=False)
            a = bk.HMCDiag(model_of(), 0.11, 5, path="step", prefetch_rng=False, **kw)
            b = bk.HMCDiag(model_of(), 0.11, 5, path="auto", prefetch_rng=prefetch, **kw)
            assert b._fused_draw and b._fused_zt == (D >= 32) and not a._fused
            for n in range(6):
                ta, la = a.sample()
                tb, lb = b.sample()
                assert torch.equal(ta, tb) and torch.equal(la, lb), (C, D, prefetch, n)
                assert torch.equal(a.last_accept, b.last_accept)
            np.testing.assert_array_equal(a.rng_state(), b.rng_state())
            assert a.accept_rate() == b.accept_rate()


def test_hmc_whole_draw_kernel_energies_are_the_reduction_kernels_values(ops):
    """kin0 / kin1 / lp of bk_hmc_draw_gaussian == bk_leapfrog_finish / bk_target_diag_gaussian_grad
    on the same arrays, bit for bit (same quarter-wise summation order), for both momentum layouts."""
    rng = np.random.default_rng(9)
    for C, D in [(70, 37), (256, 256), (33, 1001)]:
        dev = ops.device
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        th, lam, met = t(rng.normal(size=(D, C))), t(np.logspace(0, 1, D)), t(rng.uniform(0.5, 2, size=D))
        dp = (D + 7) // 8 * 8
        zt = t(rng.normal(size=(C, dp)))
        rho0 = zt[:, :D].t().contiguous()
        eps, L = 0.07, 4
        th_ref, rho_ref = torch.empty_like(th), torch.empty_like(th)
        ops.hmc_trajectory_gaussian(th, th_ref, rho0, rho_ref, lam, met, eps, L)
        k0_ref, k1_ref, lp_ref = (torch.empty(C, dtype=torch.float64, device=dev) for _ in range(3))
        ops.leapfrog_finish(rho0, None, None, met, 0.0, False, k0_ref)
        ops.leapfrog_finish(rho_ref, None, None, met, 0.0, False, k1_ref)
        ops.target_grad("diag_gaussian", lam, th_ref, None, lp_ref)
        for use_zt in (True, False):
            out = torch.empty_like(th)
            part = torch.empty(12 * C, dtype=torch.float64, device=dev)
            k0, k1, lp = (torch.empty(C, dtype=torch.float64, device=dev) for _ in range(3))
            ops.hmc_draw_gaussian(th, out, None if use_zt else rho0, zt if use_zt else None, lam, met, eps, L, part, k0, k1, lp)
            assert torch.equal(out, th_ref) and torch.equal(k0, k0_ref) and torch.equal(k1, k1_ref) and torch.equal(lp, lp_ref)
            # the accept test folded into the same launches == bk_mh_accept on those energies
            lp_cur = (lp_ref - k1_ref + k0_ref) + t(rng.normal(size=C))  # energy errors of order one
            logu = t(np.log(rng.uniform(size=C)))
            f64 = dict(dtype=torch.float64, device=dev)
            la, lb = lp_cur.clone(), lp_cur.clone()
            ma, mb = (torch.zeros(C, dtype=torch.uint8, device=dev) for _ in range(2))
            ra, rb = torch.empty(C, **f64), torch.empty(C, **f64)
            na, nb = (torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2))
            ops.mh_accept(0, la, k0_ref, lp_ref, k1_ref, logu, ma, ra, na)
            ops.hmc_draw_gaussian(th, out, None if use_zt else rho0, zt if use_zt else None, lam, met, eps, L, part, k0, k1, lp,
                                  accept=(lb, logu, mb, rb, nb))
            assert torch.equal(ma, mb) and torch.equal(ra, rb) and torch.equal(la, lb) and int(na) == int(nb)
            assert 0 < int(nb) < C


@pytest.mark.parametrize("D,K", [(11, 3), (101, 3), (21, 2), (129, 3), (40, 4)])
def test_drghmc_device_side_lane_counts_equal_host_sized_launches(ops, D, K):
    """Lane counts kept on the device (every launch sized for the parent set, no host read, the
    draw replayed as one hipGraph) against launches sized by reading the counts back: same draws,
    same joint log densities, same momenta, same stream positions, same trajectories run."""
    sizes, counts = [0.3, 0.1, 0.03, 0.01][:K], [3, 6, 12, 24][:K]
    for C in (700, 64):
        mk = lambda **kw: bk.DrGhmcDiag(bk.Funnel(D), K, sizes, counts, 0.3, chains=C, seed=5, **kw)  # noqa: E731
        a = mk(device_counts=False)
        b = mk(device_counts=True, graph=False)
        g = mk()  # default: device counts + hipGraph replay
        u = mk(graph=False, fuse_first_ghost=False, recompute_gradient=False)  # every ghost a launch of its own, cached gradients
        assert g._dev_counts and g._use_graph and not a._dev_counts and g.host_syncs_per_draw == 0
        assert g._regrad and b._regrad and not u._regrad and not a._regrad
        seen = set()
        for n in range(12):
            ta, la = a.sample()
            tb, lb = b.sample()
            tg, lg = g.sample()
            tu, lu = u.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (D, K, C, n)
            assert torch.equal(ta, tg) and torch.equal(la, lg), (D, K, C, n, "graph")
            assert torch.equal(ta, tu) and torch.equal(la, lu), (D, K, C, n, "unfused ghosts")
            assert a.last_stage_lanes == b.last_stage_lanes == g.last_stage_lanes == u.last_stage_lanes
            assert a.last_lane_steps == b.last_lane_steps == g.last_lane_steps
            assert a.last_grad_evals == b.last_grad_evals == g.last_grad_evals
            seen.update(t for t, _ in a.last_stage_lanes)
        assert torch.equal(a._rho, b._rho) and torch.equal(a._rho, g._rho)
        np.testing.assert_array_equal(a.rng_state(), b.rng_state())
        np.testing.assert_array_equal(a.rng_state(), g.rng_state())
        assert g._graph is not None and len(seen) >= 3
        total = float(g.lane_steps_total.item())
        assert total > 0 and total == float(b.lane_steps_total.item())
        # the one-launch path keeps no gradient cache (recompute_gradient): its checkpoint carries an up-to-date one all the
        # same, and a sampler on the host-sized path resumes from it with the same draws
        c = mk(device_counts=False)
        c.load_state_dict(g.state_dict())
        assert torch.equal(c._grad, a._grad) and torch.equal(c._lp, a._lp)
        for n in range(3):
            tg, lg = g.sample()
            tc, lc = c.sample()
            ta, la = a.sample()
            assert torch.equal(tg, tc) and torch.equal(lg, lc) and torch.equal(ta, tc), (D, K, C, n, "resumed on another path")
        np.testing.assert_array_equal(g.rng_state(), c.rng_state())


def funnel_plugin(D):
    """examples/plugin_target/libfunnel_target.so: a user target exporting bk_target_fn AND bk_target_fn_n."""
    return bk.CTarget(os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so"), "funnel_target", D,
                      counted_symbol="funnel_target_n")


@pytest.mark.parametrize("D,K", [(11, 3), (101, 3), (21, 2), (40, 4), (150, 2)])
def test_drghmc_model_opaque_device_counts_equal_host_sized_and_one_launch(ops, D, K):
    """the gradient as a SEPARATE op per leapfrog step (drghmc.py:280-283) -- the library's
    counted funnel op, and a user plugin behind bk_target_fn_n -- with every lane count on the device: no host
    read in a draw, one hipGraph; same draws, joint log densities, momenta, stream positions and trajectories as
    the host-sized compacted path and as the one-launch proposal kernel."""
    sizes, counts = [0.3, 0.1, 0.03, 0.01][:K], [3, 6, 12, 24][:K]
    for C in (700, 64):
        mk = lambda model, **kw: bk.DrGhmcDiag(model, K, sizes, counts, 0.3, chains=C, seed=5, **kw)  # noqa: E731
        a = mk(bk.Funnel(D), device_counts=False, path="opaque")  # host-sized, gradient op per step
        a1 = mk(bk.Funnel(D), device_counts=False, path="step")  # host-sized, {gradient, kick, drift} one launch
        e0 = mk(bk.Funnel(D), path="opaque")     # counted, gradient op per step, one hipGraph
        f = mk(bk.Funnel(D)) if D <= 129 else None                       # one launch per proposal (+ graph)
        e = mk(bk.Funnel(D), path="step", graph=False)           # counted steps, eager
        g = mk(bk.Funnel(D), path="step")                        # counted steps, one hipGraph (the default)
        p = mk(funnel_plugin(D))                                        # the user's plugin, counted, one hipGraph
        h = mk(funnel_plugin(D), device_counts=False)                   # the plugin, host-sized
        for s_ in (e, g, p, e0):
            assert s_._dev_counts and not s_._one_launch and s_.host_syncs_per_draw == 0
        assert g._step_hook and e._step_hook and a1._step_hook and not a._step_hook and not e0._step_hook and not p._step_hook
        assert g._use_graph and p._use_graph and not e._use_graph and not a._dev_counts and not h._dev_counts
        assert f is None or f._one_launch
        # (the library's funnel and the plugin sum the coordinates in the same canonical class order for every D)
        same_order = True
        for n in range(12):
            ta, la = a.sample()
            th_, lh_ = h.sample()
            for name, s_ in (("one launch", f), ("eager", e), ("graph", g), ("plugin", p), ("plugin host-sized", h),
                             ("host-sized one-launch steps", a1), ("counted gradient op", e0)):
                if s_ is None:
                    continue
                if s_ is h:
                    t_, l_ = th_, lh_
                else:
                    t_, l_ = s_.sample()
                if not same_order and s_ in (p, h):
                    assert torch.equal(th_, t_) and torch.equal(lh_, l_), (D, K, C, n, name)
                    assert h.last_stage_lanes == s_.last_stage_lanes, (D, K, C, n, name)
                    continue
                assert torch.equal(ta, t_), (D, K, C, n, name)
                if s_ is f:
                    # (the proposal kernel sums a trajectory's kinetic energy in its own lanes' order, bk_leapfrog_finish
                    # in four quarters: the joint log densities agree to rounding, the draws bit for bit)
                    torch.testing.assert_close(la, l_, rtol=1e-13, atol=1e-13)
                else:
                    assert torch.equal(la, l_), (D, K, C, n, name)
                assert a.last_stage_lanes == s_.last_stage_lanes, (D, K, C, n, name)
            assert a.last_lane_steps == g.last_lane_steps and h.last_lane_steps == p.last_lane_steps
            assert a.last_grad_evals == g.last_grad_evals
        for s_ in (f, e, g, p, h):
            if s_ is not None:
                ref_ = a if (same_order or s_ not in (p, h)) else h
                assert torch.equal(ref_._rho, s_._rho)
                np.testing.assert_array_equal(ref_.rng_state(), s_.rng_state())
        assert g._graph is not None and p._graph is not None
        assert float(g.lane_steps_total.item()) == float(e.lane_steps_total.item()) > 0


def test_drghmc_counted_steps_on_gaussians_with_metric(ops):
    """The counted step-by-step draw on the separable Gaussians (with a diagonal metric, without probabilistic retry) --
    gradient a separate op, and {gradient, kick, drift} one launch per step -- against the host-sized path, bit for bit; and
    (round 5: a separable density is a lanes-form density without head coordinates) the one-launch proposals: same draws, the
    joint log density to rounding (summed in the lanes' order); odd and even chain counts; D past 128 has no one-launch kernel."""
    for C, D in ((257, 16), (1024, 33), (130, 200)):
        lam, met = np.linspace(0.5, 3.0, D), np.linspace(0.8, 1.3, D)
        for model, kw in ((lambda: bk.DiagGaussian(lam), dict(metric_diag=met)), (lambda: bk.IsoGaussian(D), dict(prob_retry=False))):
            mk = lambda **k2: bk.DrGhmcDiag(model(), 3, [0.9, 0.4, 0.15], [2, 4, 6], 0.5, chains=C, seed=3, **kw, **k2)  # noqa: E731
            a = mk(device_counts=False, path="opaque")
            g = mk(path="opaque")
            h = mk(path="step")
            f = mk()
            assert g._dev_counts and g._use_graph and not g._one_launch and not a._dev_counts and h._step_hook and not g._step_hook
            assert f._one_launch == (D <= 128) and f._dev_counts
            for n in range(10):
                ta, la = a.sample()
                tg, lg = g.sample()
                th_, lh = h.sample()
                tf, lf = f.sample()
                assert torch.equal(ta, tg) and torch.equal(la, lg), (C, D, n)
                assert torch.equal(ta, th_) and torch.equal(la, lh), (C, D, n)
                assert torch.equal(ta, tf), (C, D, n)
                torch.testing.assert_close(la, lf, rtol=1e-12, atol=1e-12)
                assert a.last_stage_lanes == g.last_stage_lanes == f.last_stage_lanes
            assert torch.equal(a._rho, g._rho) and torch.equal(a._rho, f._rho)
            np.testing.assert_array_equal(a.rng_state(), g.rng_state())
            np.testing.assert_array_equal(a.rng_state(), f.rng_state())


def test_checkpoint_of_sampler_moments_recorder_and_draw_store(ops, tmp_path):
    """SURVEY 8f.4 end to end on the device: sampler + Welford moments + tracked series + chunked
    [draw, D, C] store, checkpointed at draw 20 of 40 and restored into fresh objects."""
    from tests.diag_parity import check_checkpoint_of_sampler_and_diagnostics

    check_checkpoint_of_sampler_and_diagnostics(ops, str(tmp_path), chains=300, D=24, draws=40, at=20)


def test_rhat_and_ess_of_device_chains_match_the_cpu_reference_pipeline(ops):
    """north_star: "R-hat / ESS within 1 % of the CPU reference".  HIP sampler -> RunningMoments /
    DrawRecorder against oracle sampler -> oracle/diagnostics.py on the same seeds.
    (a) DiagGaussian D=16, 256 chains x 400 draws: the chains are bit-identical, so the diagnostics
        agree to summation order (R-hat rel 1e-9, ESS rel 1e-9 -- far inside 1 %).
    (b) Neal's funnel D=11, DRGHMC K=3, 96 chains x 300 draws: chains agree to the funnel tolerance
        (chaotic flow; 60 draws), the diagnostics agree accordingly: R-hat rel 1e-6, ESS summed over
        chains within 1 % (the estimator's truncation point can flip for a single chain)."""
    from oracle import diagnostics as od
    from oracle import models as om
    from oracle import samplers as osamp

    # (a)
    C, D, N, seed = 256, 16, 400, 4242
    lam = np.logspace(0, 1, D)
    s = bk.HMCDiag(bk.DiagGaussian(lam), 0.25, 6, chains=C, seed=seed)
    mom = bk.RunningMoments(D, C)
    rec = bk.DrawRecorder([0, 7, D - 1], N, C)
    for _ in range(N):
        th, lp = s.sample()
        mom.update(th)
        rec.record(th, lp)
    ref = np.empty((N, C, D))
    ref_lp = np.empty((N, C))
    for c in range(C):
        o = osamp.HMCDiag(om.DiagGaussian(lam), 0.25, 6, seed=np.random.Philox(key=[seed, c]))
        for n in range(N):
            ref[n, c], ref_lp[n, c] = o.sample()
    assert np.array_equal(rec.series[0, :N].cpu().numpy(), ref[:, :, 0])  # the chains themselves: bit-identical
    want_rhat = np.array([od.rhat([ref[:, c, d] for c in range(C)]) for d in range(D)])
    np.testing.assert_allclose(mom.rhat(), want_rhat, rtol=1e-9)
    np.testing.assert_allclose(rec.rhat()[:3], want_rhat[[0, 7, D - 1]], rtol=1e-9)
    ess_dev = rec.ess().cpu().numpy()
    for k, d in enumerate([0, 7, D - 1]):
        want = np.array([od.ess(ref[:, c, d]) for c in range(C)])
        np.testing.assert_allclose(ess_dev[k], want, rtol=1e-9)
    np.testing.assert_allclose(ess_dev[3], [od.ess(ref_lp[:, c]) for c in range(C)], rtol=1e-8)  # joint logp: a reduction
    # (b)  60 draws: the regime in which the device chains still track the CPU chains (funnel
    # tolerance), so the diagnostics are functions of (nearly) the same series
    C, D, N, seed = 256, 11, 60, 777
    args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1)
    s = bk.DrGhmcDiag(bk.Funnel(D), *args, chains=C, seed=seed)
    mom = bk.RunningMoments(D, C)
    rec = bk.DrawRecorder([0, 1, D - 1], N, C)
    for _ in range(N):
        th, lp = s.sample()
        mom.update(th)
        rec.record(th, lp)
    ref = np.empty((N, C, D))
    for c in range(C):
        o = osamp.DrGhmcDiag(om.Funnel(D), *args, seed=np.random.Philox(key=[seed, c]))
        for n in range(N):
            ref[n, c] = o.sample()[0]
    got = rec.series[:3, :N].cpu().numpy()
    for n in range(45):  # (256 chains: the fastest-diverging chain outruns the schedule near draw 60)
        np.testing.assert_allclose(got[0, n], ref[n, :, 0], **funnel_tol(n))
    np.testing.assert_allclose(got[0], ref[:, :, 0], rtol=1e-5, atol=1e-6)
    want_rhat = np.array([od.rhat([ref[:, c, d] for c in range(C)]) for d in range(D)])
    rh = mom.rhat()
    print("funnel R-hat device vs CPU pipeline, max rel diff:", np.abs(rh / want_rhat - 1).max())
    np.testing.assert_allclose(rh, want_rhat, rtol=1e-6)
    ess_dev = rec.ess().cpu().numpy()
    for k, d in enumerate([0, 1, D - 1]):
        want = np.array([od.ess(ref[:, c, d]) for c in range(C)])
        tot_dev, tot_ref = np.clip(ess_dev[k], 0, N).sum(), np.clip(want, 0, N).sum()
        print(f"funnel ESS of theta[{d}] summed over chains: device {tot_dev:.3f} CPU {tot_ref:.3f}")
        assert abs(tot_dev / tot_ref - 1) <= 1e-2
        assert np.mean(np.abs(ess_dev[k] - want) <= 1e-6 * np.abs(want)) > 0.98  # chain by chain, but for truncation flips


def test_full_size_cfg5_properties(ops):
    """BASELINE.json config 5 at its real size on one GPU (no reference oracle: properties).  Logistic
    regression N = 1e6, D = 512, 2,048 chains: gradient finite and equal to torch's fp64 matmul on a
    slice of chains (<= 1e-12 relative), shard invariance (a block of chains evaluated alone gives the
    same log densities and gradients bit for bit), one HMC draw with the dense metric,
    one annealed-SMC temperature."""
    N, D, C = 1_000_000, 512, 2048
    dev = ops.device
    g = torch.Generator(device=dev)
    g.manual_seed(20243)
    X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=g) / D ** 0.5
    tstar = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
    y = (torch.rand(N, dtype=torch.float64, device=dev, generator=g) < torch.sigmoid(X @ tstar)).to(torch.float64)
    model = bk.LogisticRegression(X, y, prior_scale=1.0)
    th = torch.randn((D, C), dtype=torch.float64, device=dev, generator=g) * 0.1
    grad, lp = torch.empty_like(th), torch.empty(C, dtype=torch.float64, device=dev)
    model.bk_eval(th, grad, lp)
    assert torch.isfinite(grad).all() and torch.isfinite(lp).all()
    z = X @ th[:, :16]
    r = y[:, None] - torch.sigmoid(z)
    gref = X.t() @ r - th[:, :16]
    lref = (y[:, None] * z - torch.nn.functional.softplus(z)).sum(dim=0) - 0.5 * (th[:, :16] ** 2).sum(dim=0)
    assert float(((grad[:, :16] - gref).abs().max() / gref.abs().max()).item()) <= 1e-12
    np.testing.assert_allclose(lp[:16].cpu().numpy(), lref.cpu().numpy(), rtol=1e-11)
    # a gradient-only call (what a leapfrog step makes: no log likelihood formed) gives the same gradient, bit for bit;
    # N = 1e6 is 7,812 whole 128-row blocks + 64 rows: the last rows go through the checked GEMM launch
    g_only = torch.full_like(grad, float("nan"))
    model.bk_eval(th, g_only, None)
    assert torch.equal(g_only, grad)
    # shard invariance: chains [512, 1024) alone
    # (as views with the full arrays' leading dimension: the target's scratch is laid out for C chains)
    g2 = torch.full((D, C), float("nan"), dtype=torch.float64, device=dev)[:, :512]
    l2 = torch.empty(512, dtype=torch.float64, device=dev)
    model.bk_eval(th[:, 512:1024], g2, l2)
    # the split of the N = 1e6 contraction is a function of (D, N) only: a shard reproduces its columns bit for bit
    assert torch.equal(g2, grad[:, 512:1024])
    assert torch.equal(l2, lp[512:1024])  # (the log-likelihood partials are summed per chain in a fixed order)
    # one dense-metric HMC draw and one SMC temperature at full size
    Md = torch.eye(D, dtype=torch.float64) * (4.0 * D / N)
    s = bk.HMCDiag(model, 0.3, 2, chains=C, seed=20243, metric_dense=Md, init=th.t().contiguous().cpu())
    t1, l1 = s.sample()
    assert torch.isfinite(t1).all() and torch.isfinite(l1).all() and 0.0 <= s.accept_rate() <= 1.0
    init = torch.randn((C, D), dtype=torch.float64, device=dev, generator=g)
    # the reference's ladder t = n / N (smc.py:42-43) at this size: the first reweighting of any affordable N leaves
    # one or two particles (log likelihoods of prior draws differ by thousands) ...
    smc = bk.TemperedLikelihoodSMC(model, C, 8, init, bk.hmc_kernel(0.5, 1, metric_dense=Md), seed=20243)
    smc.transition(1)
    assert torch.isfinite(smc.thetas).all() and 1.0 <= smc.last_ess < 0.01 * C
    # ... the adaptive ladder takes the steps the problem allows: three REAL steps (move, reweight, resample) at full
    # size, each keeping half the particles, temperatures of the order 1e-4 (tools/config5_ladder.py runs all 127)
    ad = bk.TemperedLikelihoodSMC(model, C, 8, init, bk.hmc_kernel(0.35, 2, adapt_metric=True), seed=20243, adaptive=0.5)
    for n in (1, 2, 3):
        ad.transition(n)
        assert ad.last_ess >= C / 4 and ad.last_ess >= 0.5 * C * (1 - 1e-3), (n, ad.last_ess)
    assert 0.0 < ad.temperatures[0] < ad.temperatures[1] < ad.temperatures[2] < 0.01
    assert torch.isfinite(ad.thetas).all() and min(ad.kernel.accept_rates) > 0.3


def test_scalars_assigned_between_draws_reach_a_replayed_graph(ops):
    """Step sizes, step counts and damping are plain attributes in the reference (hmc.py:18-19, mala.py:23,
    drghmc.py:60-66): a step-size adaptation assigns them between draws.  A captured hipGraph bakes them into
    its launches, so they are compared before every replay and a change -- assignment or an in-place edit of
    the list -- captures again.  Graph samplers must follow the eager ones bit for bit through such edits."""
    lam = np.logspace(0, 1, 16)
    a = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=512, seed=3, graph=True)
    b = bk.HMCDiag(bk.DiagGaussian(lam), 0.05, 5, chains=512, seed=3, graph=False, prefetch_rng=False)
    ma = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=512, seed=4, graph=True)
    mb = bk.MALA(bk.DiagGaussian(lam), 0.02, chains=512, seed=4, graph=False, prefetch_rng=False)
    da = bk.DrGhmcDiag(bk.Funnel(21), 3, [0.3, 0.1, 0.03], [3, 6, 12], 0.3, chains=700, seed=5)  # device counts + graph
    db = bk.DrGhmcDiag(bk.Funnel(21), 3, [0.3, 0.1, 0.03], [3, 6, 12], 0.3, chains=700, seed=5, device_counts=False)
    assert da._use_graph and not db._use_graph
    for n in range(12):
        if n == 4:
            a._stepsize = b._stepsize = 0.07
            ma._epsilon = mb._epsilon = 0.03
            for d in (da, db):
                d._damping = 0.5
                d._leapfrog_step_sizes[0] = 0.25   # in place: no assignment to intercept
        if n == 8:
            a._steps = b._steps = 3
            for d in (da, db):
                d._leapfrog_step_counts = [2, 6, 10]
        for x, y in ((a, b), (ma, mb), (da, db)):
            tx, lx = x.sample()
            ty, ly = y.sample()
            assert torch.equal(tx, ty) and torch.equal(lx, ly), (type(x).__name__, n)
        assert da.last_stage_lanes == db.last_stage_lanes and da.last_lane_steps == db.last_lane_steps, n
    assert a._graph is not None and ma._graph is not None and da._graph is not None
    np.testing.assert_array_equal(da.rng_state(), db.rng_state())


def test_drghmc_attached_diagnostics_equal_manual_updates(ops):
    """attach(): Welford moments and tracked series updated by launches INSIDE the replayed hipGraph of a draw
    (update count / row index read from the sampler's device-side draw counter) == update() / record() after
    every sample(); advance() draws without returned copies."""
    from tests.sampler_parity import check_attached_diagnostics

    g = check_attached_diagnostics(ops, C=700, D=21, draws=15)
    assert g._use_graph and g._graph is not None
    check_attached_diagnostics(ops, C=700, D=21, draws=8, graph=False)
    check_attached_diagnostics(ops, C=130, D=11, draws=8, device_counts=False)
    check_attached_diagnostics(ops, C=4096, D=21, draws=6)  # padded rows: theta's row pitch differs from the moments'


@pytest.mark.parametrize("D,C", [(21, 900), (40, 901), (101, 4096)])
def test_drghmc_advance_n_replays_graphs_of_several_draws(ops, D, C):
    """advance(n): hipGraphs of up to DRAWS_PER_GRAPH consecutive draws (one graph launch instead of n) leave exactly what n
    advance() calls leave -- state, momenta, stream positions, lane statistics, the attached moments and series.  Inside such a
    graph the moments update of a draw is a job of the next draw's generator launch (D = 101: padded state rows) or, where that
    launch cannot carry it, a launch of its own ahead of it (D = 21: no wavefront-per-chain generator; C = 901: odd)."""
    K = 3
    mk = lambda: bk.DrGhmcDiag(bk.Funnel(D), K, [0.3, 0.1, 0.03], [3, 6, 12], 0.3, chains=C, seed=9)  # noqa: E731
    a, b = mk(), mk()
    ma, mb = bk.RunningMoments(D, C), bk.RunningMoments(D, C)
    ra, rb = bk.DrawRecorder([0, D - 1], 64, C), bk.DrawRecorder([0, D - 1], 64, C)
    a.attach(moments=ma, recorder=ra)
    b.attach(moments=mb, recorder=rb)
    for _ in range(37):
        a.advance()
    b.advance(37)          # 2 warm-up draws one by one, then graphs of 10, 10, 10 and 5 draws
    assert b._graph_many and set(b._graph_many) <= {10, 5} and a._draws == b._draws == 37 and ma.n == mb.n == 37 and ra.n == rb.n == 37
    assert torch.equal(a._theta_dc, b._theta_dc) and torch.equal(a._rho, b._rho) and torch.equal(a._cur_H, b._cur_H)
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    assert a.last_stage_lanes == b.last_stage_lanes and float(a.lane_steps_total.item()) == float(b.lane_steps_total.item())
    assert torch.equal(torch.as_tensor(ma.rhat()), torch.as_tensor(mb.rhat())) and torch.equal(ra.series[:, :37], rb.series[:, :37])
    # a changed step size is seen by the next call (the graphs bake it in), and sample() continues the same chain of draws
    a._leapfrog_step_sizes = [0.25, 0.1, 0.03]
    b._leapfrog_step_sizes = [0.25, 0.1, 0.03]
    for _ in range(12):
        a.advance()
    b.advance(12)
    ta, la = a.sample()
    tb, lb = b.sample()
    assert torch.equal(ta, tb) and torch.equal(la, lb) and ra.n == rb.n == 50
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    with pytest.raises(IndexError):
        b.advance(20)      # the recorder holds 64 draws


def test_recorder_dims_square_draws_padded_moments_and_attach_after_restore(ops):
    from tests.sampler_parity import check_recorder_and_moments_edges

    check_recorder_and_moments_edges(ops)
    check_recorder_and_moments_edges(ops, C=130, D=9)


def test_adaptive_smc_ladder_keeps_its_ess_and_finds_the_posterior(ops):
    from tests.sampler_parity import check_adaptive_smc_ladder

    check_adaptive_smc_ladder(ops, M=4096, D=6, n_obs=100_000)


def test_annealed_smc_on_the_logistic_target_against_a_long_hmc_run(ops):
    """config 5's algorithm at reduced size (20,000 observations, 16 coefficients, 2,048
    particles) -- the adaptive ladder with HMC moves under the particle-adapted dense metric -- against a long
    many-chain HMC run on the same posterior: every coefficient's posterior mean within 4 Monte-Carlo standard errors,
    posterior standard deviations within 15 %, every reweighting keeping half the particles."""
    N, D, M = 20_000, 16, 2048
    dev = ops.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=g) / D ** 0.5
    tstar = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
    y = (torch.rand(N, dtype=torch.float64, device=dev, generator=g) < torch.sigmoid(X @ tstar)).to(torch.float64)
    model = bk.LogisticRegression(X, y, prior_scale=1.0)
    init = torch.randn((M, D), dtype=torch.float64, device=dev, generator=g)
    smc = bk.TemperedLikelihoodSMC(model, M, 1, init, bk.hmc_kernel(0.6, 3, adapt_metric=True), seed=11, adaptive=0.5)
    smc.run()
    assert smc.temperatures[-1] == 1.0 and 10 < len(smc.temperatures) < 300
    assert min(smc.ess_history) >= 0.5 * M * (1 - 1e-3) and min(smc.kernel.accept_rates) > 0.4
    P = smc.thetas
    m_smc, v_smc = P.mean(dim=0), P.var(dim=0)
    # the long run: 512 chains from the SMC's particles (already in the posterior), dense metric = their covariance
    C, warm, draws = 512, 50, 400
    cov = torch.cov(P.t()).cpu()
    h = bk.HMCDiag(model, 0.5, 4, chains=C, seed=12, metric_dense=cov, init=P[:C].contiguous().cpu())
    for _ in range(warm):
        h.sample()
    acc = torch.zeros((C, D), dtype=torch.float64, device=dev)
    acc2 = torch.zeros_like(acc)
    for _ in range(draws):
        th, _ = h.sample()
        acc += th
        acc2 += th * th
    assert 0.5 < h.accept_rate() <= 1.0
    chain_means = acc / draws
    m_hmc = chain_means.mean(dim=0)
    mcse_hmc = chain_means.std(dim=0) / C ** 0.5           # between-chain spread of the chain means
    v_hmc = (acc2.sum(dim=0) / (C * draws)) - m_hmc * m_hmc
    # the SMC's own Monte-Carlo error: half the particles are distinct after the last resampling, a quarter counted
    mcse_smc = (v_hmc / (M / 4)) ** 0.5
    z = (m_smc - m_hmc) / (mcse_hmc ** 2 + mcse_smc ** 2) ** 0.5
    assert float(z.abs().max().item()) < 4.0, z
    assert float((v_smc.sqrt() / v_hmc.sqrt() - 1.0).abs().max().item()) < 0.15


def test_single_chain_one_launch_per_draw_equals_the_step_by_step_path(ops):
    """the single-chain drop-in (a reference-style NumPy model, the README example) with ONE launch
    per MALA draw (bk_mala_single_draw: proposal densities, accept test, select and the next proposal; model outputs
    and draws through pinned, device-addressable host memory) against the step-by-step launches: same draws, same
    returned log densities, same accept flags, same stream position after every draw -- PCG64 (an int seed: the
    reference's own stream) and Philox, D = 1 and D = 37; one model call per draw + one at construction."""
    class StdNormal:  # README.md:17-23
        def dims(self):
            return 1

        def log_density(self, theta):
            return -0.5 * theta[0] * theta[0]

        def log_density_gradient(self, theta):
            return -0.5 * theta[0] * theta[0], -theta

    class Gauss:
        def __init__(self, lam):
            self.lam, self.calls = np.asarray(lam, dtype=np.float64), 0

        def dims(self):
            return self.lam.shape[0]

        def log_density(self, th):
            return float(-0.5 * np.dot(th, self.lam * th))

        def log_density_gradient(self, th):
            self.calls += 1
            return float(-0.5 * np.dot(th, self.lam * th)), -(self.lam * th)

    for make, eps, seed in ((lambda: StdNormal(), 0.2, 12345), (lambda: Gauss(np.linspace(0.5, 4.0, 37)), 0.05, np.random.Philox(key=[7, 3])),
                            (lambda: Gauss([2.0]), 0.3, 99)):
        mk_seed = (lambda: np.random.Philox(key=[7, 3])) if not isinstance(seed, int) else (lambda: seed)
        ma, mb = make(), make()
        a = bk.MALA(ma, eps, seed=mk_seed(), single_launch=False)
        b = bk.MALA(mb, eps, seed=mk_seed())
        assert b._single and not a._single and "one launch" in b.path
        for n in range(60):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert np.array_equal(ta, tb) and la == lb and type(lb) is np.float64, n
            assert a.last_accept == b.last_accept
            if n % 17 == 0:
                np.testing.assert_array_equal(a.rng_state(), b.rng_state())
                assert np.array_equal(a._theta, b._theta) and a._log_p_theta == b._log_p_theta
        np.testing.assert_array_equal(a.rng_state(), b.rng_state())
        assert abs(a.accept_rate() - b.accept_rate()) < 1e-12
        if hasattr(mb, "calls"):
            assert mb.calls == 60 + 1   # mala.py:31 once, :46 once per draw
        # checkpoint in the middle of the pipeline (a proposal is pending), restore into a fresh sampler, go on
        sd = b.state_dict()
        c = bk.MALA(make(), eps, seed=mk_seed())
        c.load_state_dict(sd)
        for n in range(10):
            tb, lb = b.sample()
            tc, lc = c.sample()
            assert np.array_equal(tb, tc) and lb == lc, n
        # refresh_cache() discards the pending proposal and draws its normals again: the stream does not move
        before = b.rng_state().copy()
        b.refresh_cache()
        np.testing.assert_array_equal(b.rng_state(), before)
        tb, _ = b.sample()
        tc, _ = c.sample()
        assert np.array_equal(tb, tc)


def test_logistic_retemper_equals_a_fresh_evaluation(ops):
    from tests.sampler_parity import check_logistic_retemper

    check_logistic_retemper(ops)
    check_logistic_retemper(ops, N=40_000, D=64, C=300)


def test_single_chain_one_launch_mala_against_the_oracle_over_random_shapes(ops):
    """The single-chain drop-in (one launch per draw) against the oracle's MALA -- the reference's arithmetic on the
    reference's NumPy streams -- over random dimensions, step sizes and seeds of both stream kinds: draws bit for bit,
    the returned log density to summation order, the generator's position exact."""
    from oracle import samplers as osamp

    class Gauss:
        def __init__(self, lam):
            self.lam = np.asarray(lam, dtype=np.float64)

        def dims(self):
            return self.lam.shape[0]

        def log_density(self, th):
            return -0.5 * float(np.sum(th * (self.lam * th)))

        def log_density_gradient(self, th):
            return -0.5 * float(np.sum(th * (self.lam * th))), -(self.lam * th)

    rng = np.random.default_rng(2024)
    for trial in range(12):
        D = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 70, 200]))
        lam = rng.uniform(0.5, 4.0, size=D)
        eps = float(rng.uniform(0.02, 0.4)) / max(1.0, D ** 0.33)
        if trial % 2:
            mk_seed = lambda k=int(rng.integers(1, 2 ** 31)): k                       # noqa: E731  int -> PCG64, the reference's default_rng
        else:
            key = [int(rng.integers(1, 2 ** 40)), int(rng.integers(0, 1000))]
            mk_seed = lambda key=key: np.random.Philox(key=key)                       # noqa: E731
        s = bk.MALA(Gauss(lam), eps, seed=mk_seed())
        o = osamp.MALA(Gauss(lam), eps, seed=mk_seed())
        assert s._single
        np.testing.assert_array_equal(s._theta, o._theta)
        for n in range(25):
            th, lp = s.sample()
            oth, olp = o.sample()
            assert np.array_equal(th, oth), (trial, D, n)
            assert abs(lp - olp) <= 1e-12 * max(1.0, abs(olp)), (trial, D, n)
        from tests.helpers import rng_state_words

        w = rng_state_words(o._rng)   # (11 words for Philox, 4 for PCG64)
        np.testing.assert_array_equal(s.rng_state()[:len(w), 0], w)
