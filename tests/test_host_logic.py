"""Host-side control flow of the samplers on CPU, through an injected fake ops object.

These tests do NOT exercise the HIP kernels (tests/test_gpu_*.py do, on the GPU box); they
check that the Python drivers issue the right sequence of device operations, bridge models
correctly and mirror the reference's call contracts, against the golden vectors.
"""
import functools

import numpy as np
import pytest

import bayes_kit_amd as bk
from tests.fake_ops import FakeOps
from tests.sampler_parity import check_many_chain, check_single_chain_host_model

MANY = ["hmc_stdnormal", "hmc_steps0", "hmc_iso4", "hmc_diag16_metric", "mala_stdnormal", "mala_iso8",
        "mala_diag16", "mala_diag48", "mala_init", "drghmc_stdnormal_k3", "drghmc_iso4_k2_noretry", "drghmc_k1",
        "drghmc_funnel11_k3", "drghmc_funnel101_cfg4", "drghmc_diag16_metric", "drghmc_diag40",
        "drghmc_funnel17_k4", "drghmc_funnel33_k2_metric_noretry", "metropolis_rw_iso3", "mh_ar_iso2",
        "drghmc_funnel130_k2", "drghmc_iso64_k3_damp1", "hmc_diag40_metric_steps1"]
SINGLE = ["hmc_pcg_seed", "hmc_iso4", "mala_stdnormal", "mala_init", "drghmc_stdnormal_k3", "drghmc_k1",
          "hmc_ref_binomial", "mala_ref_binomial", "drghmc_ref_binomial",
          "metropolis_rw_iso3", "mh_ar_iso2", "metropolis_pcg_seed", "mala_pcg_d5"]


@pytest.mark.parametrize("name", MANY)
def test_many_chain_driver_vs_golden(name):
    s = check_many_chain(name, FakeOps())
    if name.startswith("hmc"):
        assert s._fused  # built-in Gaussians take the register-resident trajectory by default


@pytest.mark.parametrize("name", [n for n in MANY if n.startswith("hmc")])
def test_many_chain_driver_vs_golden_step_by_step(name):
    s = check_many_chain(name, FakeOps(), path="step")
    assert not s._fused


@pytest.mark.parametrize("name", SINGLE)
def test_single_chain_drop_in_vs_golden(name):
    check_single_chain_host_model(name, FakeOps(), chains=[0, 1])


def test_readme_example_cfg1():
    # BASELINE.json config 1 / README.md:13-32: MALA on StdNormal, int seed -> PCG64 stream
    check_single_chain_host_model("mala_readme_cfg1", FakeOps())


def _counter(f):
    @functools.wraps(f)
    def w(*a, **k):
        w.calls += 1
        return f(*a, **k)
    w.calls = 0
    return w


@pytest.mark.parametrize("steps", [0, 1, 10])
def test_hmc_call_count_contract(steps):
    # test/test_hmc.py:22-35: exactly 2 log_density and steps+1 gradient calls per draw
    from oracle.models import StdNormal

    model = StdNormal()
    model.log_density = _counter(model.log_density)
    model.log_density_gradient = _counter(model.log_density_gradient)
    hmc = bk.HMCDiag(model, steps=steps, stepsize=0.25, ops=FakeOps())
    hmc.sample()
    assert model.log_density.calls == 2
    assert model.log_density_gradient.calls == hmc._steps + 1


def test_init_handling():
    # test/test_theta_initialization.py:17-54
    from unittest.mock import Mock

    def mk(init, dims=1):
        m = Mock()
        m.dims = Mock(return_value=dims)
        m.log_density_gradient = Mock(return_value=(0.5, (0,)))
        m.log_density = Mock(return_value=0.5)
        del m.batched, m.bk_eval
        return [bk.HMCDiag(m, stepsize=0.25, steps=10, init=init, ops=FakeOps()),
                bk.MALA(m, epsilon=0.5, init=init, ops=FakeOps())]

    for s in mk(np.array([])):
        assert s._theta.shape == (1,)
    for s in mk(np.array([3])):
        np.testing.assert_array_equal(s._theta, [3])
    for s in mk(np.array([3, 3, 3]), dims=3):
        np.testing.assert_array_equal(s._theta, [3, 3, 3])
        s.sample()


def test_seed_reproducibility_and_iterator():
    # test/test_hmc.py:54-65, test_mala.py:44-60, test_metropolis.py:257-263
    from oracle.models import StdNormal

    init = np.array([0.3])
    a = bk.HMCDiag(StdNormal(), steps=10, stepsize=0.25, init=init, seed=123, ops=FakeOps())
    b = bk.HMCDiag(StdNormal(), steps=10, stepsize=0.25, init=init, seed=123, ops=FakeOps())
    c = bk.HMCDiag(StdNormal(), steps=10, stepsize=0.25, init=init, seed=124, ops=FakeOps())
    da = np.array([a.sample()[0] for _ in range(25)])
    db = np.array([next(b)[0] for _ in range(25)])
    dc = np.array([draw[0] for draw, _ in zip(c, range(25))])
    np.testing.assert_array_equal(da, db)
    assert not np.array_equal(da, dc)
    assert iter(a) is a


def test_hmc_one_step_equals_mala_through_the_drivers():
    # test/test_equivalencies.py:12-32
    from oracle.models import StdNormal

    init = np.array([0.2])
    eps = 0.02
    hmc = bk.HMCDiag(StdNormal(), stepsize=eps, steps=1, init=init, seed=123, ops=FakeOps())
    mala = bk.MALA(StdNormal(), epsilon=0.5 * eps**2, init=init, seed=123, ops=FakeOps())
    d1 = np.array([hmc.sample()[0] for _ in range(50)])
    d2 = np.array([mala.sample()[0] for _ in range(50)])
    np.testing.assert_array_almost_equal(d1, d2)
    assert len(np.unique(d1)) > 20


def test_torch_autograd_model_bridge():
    import torch

    lam = torch.logspace(0, 1, 16, dtype=torch.float64)
    tm = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), 16)
    tm_dc = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), 16, layout="dc")  # fn on the (D, C) array
    ops = FakeOps()
    a = bk.HMCDiag(tm, 0.05, 8, chains=6, seed=9, ops=ops)
    b = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 8, chains=6, seed=9, ops=ops)
    c = bk.HMCDiag(tm_dc, 0.05, 8, chains=6, seed=9, ops=ops)
    for _ in range(10):
        ta, la = a.sample()
        tb, lb = b.sample()
        tc, lc = c.sample()
        np.testing.assert_allclose(ta.numpy(), tb.numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(la.numpy(), lb.numpy(), rtol=1e-11)
        np.testing.assert_allclose(tc.numpy(), tb.numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(lc.numpy(), lb.numpy(), rtol=1e-11)
    with pytest.raises(ValueError):
        bk.TorchModel(lambda Th: Th.sum(dim=1), 3, layout="rows")


def test_no_gpu_no_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle.models import StdNormal

    with pytest.raises(bk._lib.BkHipError):
        bk.HMCDiag(StdNormal(), 0.1, 3)


# ---- DRGHMC specifics ------------------------------------------------------------------------------
def _drghmc(**over):
    from oracle.models import StdNormal

    kw = dict(model=StdNormal(), max_proposals=2, leapfrog_step_sizes=[0.25, 0.25],
              leapfrog_step_counts=[1, 1], damping=0.2, ops=FakeOps())
    kw.update(over)
    return bk.DrGhmcDiag(**kw)


def test_drghmc_validation_messages():
    # error types AND texts of drghmc.py:85-207, pinned by test/test_drghmc.py:179-347
    import re

    for bad in [[1], 1.0]:
        with pytest.raises(TypeError, match=re.escape(f"max_proposals must be an int, not {type(bad)}")):
            _drghmc(max_proposals=bad)
    for bad in [0, -1]:
        with pytest.raises(ValueError, match=f"max_proposals must be greater than or equal to 1, not {bad}"):
            _drghmc(max_proposals=bad)
    for bad in [1, 0.25]:
        msg = f"leapfrog_step_sizes must be an instance of type sequence, but found type {type(bad)}"
        with pytest.raises(TypeError, match=re.escape(msg)):
            _drghmc(leapfrog_step_sizes=bad)
        msg = f"leapfrog_step_counts must be an instance of type sequence, but found type {type(bad)}"
        with pytest.raises(TypeError, match=re.escape(msg)):
            _drghmc(leapfrog_step_counts=bad)
    for bad in [[0.25], [0.25, 0.25, 0.25]]:
        msg = (f"leapfrog_step_sizes must be a sequence of length 2, so that each proposal has its own "
               f"specified leapfrog step size, but instead found length of {len(bad)}")
        with pytest.raises(ValueError, match=msg):
            _drghmc(leapfrog_step_sizes=bad)
    for bad in [[1], [1, 1, 1]]:
        msg = (f"leapfrog_step_counts must be a sequence of length 2, so that each proposal has its own "
               f"specified number of leapfrog steps, but instead found length of {len(bad)}")
        with pytest.raises(ValueError, match=msg):
            _drghmc(leapfrog_step_counts=bad)
    with pytest.raises(TypeError, match=re.escape(
            f"each step size in leapfrog_step_sizes must be of type float, but found step size of type {int} "
            f"at index 1")):
        _drghmc(leapfrog_step_sizes=[0.25, 1])
    for bad in [[-0.25, 0.25], [0.0, 0.25]]:
        with pytest.raises(ValueError, match=re.escape(
                f"each step size in leapfrog_step_sizes must be positive, but found step size of {bad[0]} at index 0")):
            _drghmc(leapfrog_step_sizes=bad)
    with pytest.raises(TypeError, match=re.escape(
            f"each step count in leapfrog_step_counts must be of type int, but found step count of type {float} "
            f"at index 1")):
        _drghmc(leapfrog_step_counts=[1, 1.0])
    for bad in [[-2, 1], [0, 1]]:
        with pytest.raises(ValueError, match=re.escape(
                f"each step count in leapfrog_step_counts must be positive, but found step count of {bad[0]} at index 0")):
            _drghmc(leapfrog_step_counts=bad)
    for bad in [[0.5], int(1)]:
        with pytest.raises(TypeError, match=re.escape(f"damping must be of type float, but found type {type(bad)}")):
            _drghmc(damping=bad)
    for bad in [float(0), float(-1)]:
        with pytest.raises(ValueError, match=re.escape(f"damping must be within (0, 1], but found damping of {bad}")):
            _drghmc(damping=bad)


def test_drghmc_gradient_call_bound_and_attrs():
    # test/test_drghmc.py:52-94: at most 1 + sum_k L_k 2^(K-1-k) gradient calls per draw
    from oracle.models import StdNormal

    model = StdNormal()
    model.log_density_gradient = _counter(model.log_density_gradient)
    counts = [2, 4, 8]
    s = bk.DrGhmcDiag(model, 3, [0.9, 0.45, 0.225], counts, 0.2, seed=5, ops=FakeOps())
    assert s._leapfrog_step_counts == counts and iter(s) is s
    bound = 1 + sum(c * 2 ** (3 - 1 - k) for k, c in enumerate(counts))
    for _ in range(60):
        before = model.log_density_gradient.calls
        next(s)
        assert model.log_density_gradient.calls - before <= bound


def test_drghmc_lane_sets_follow_the_reference_schedule():
    """Which chains run which trajectory = the union of the oracle's per-chain schedules."""
    from tests.helpers import load_case, oracle_sampler
    from tests.sampler_parity import build_sampler, product_model

    case, z = load_case("drghmc_funnel11_k3")
    C = z["draws"].shape[1]
    ops = FakeOps()
    s = build_sampler(case, product_model(case["model"], ops), ops, case["seed"], chains=C)
    oracles = [oracle_sampler(case, c) for c in range(C)]
    multi = 0
    for n in range(25):
        s.sample()
        want = {}
        for o in oracles:
            o.sample()
            for t in o.last_schedule:
                want[t] = want.get(t, 0) + 1
        got = {}
        for t, lanes in s.last_stage_lanes:
            got[t] = got.get(t, 0) + lanes
        assert got == want, (n, got, want)
        multi += len(got) > 1
    assert multi > 0


def test_diagnostics_api_vs_reference_golden():
    from tests.diag_parity import check_diagnostics

    check_diagnostics(FakeOps(), ess_rtol=1e-12)


def test_cache_tiling_is_only_a_schedule():
    ops = FakeOps()
    lam = np.logspace(0, 1, 6)
    a = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 5, chains=10, seed=5, chain_tile=0, ops=ops)
    b = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 5, chains=10, seed=5, chain_tile=4, ops=ops)
    for _ in range(5):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert np.array_equal(ta.numpy(), tb.numpy()) and np.array_equal(la.numpy(), lb.numpy())
    assert a._chain_tile == 10 and b._chain_tile == 4


def test_checkpoint_resume():
    from tests.sampler_parity import check_checkpoint_resume

    ops = FakeOps()
    lam = np.logspace(0, 1, 5)
    check_checkpoint_resume(ops, lambda: bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.1, 4, chains=6, seed=2, ops=ops))
    check_checkpoint_resume(ops, lambda: bk.MALA(bk.DiagGaussian(lam, ops=ops), 0.05, chains=6, seed=2, ops=ops))
    check_checkpoint_resume(ops, lambda: bk.DrGhmcDiag(bk.Funnel(5, ops=ops), 2, [0.3, 0.1], [3, 9], 0.3, chains=6,
                                                       seed=2, ops=ops))


def test_dense_metric_driver():
    from tests.sampler_parity import check_dense_metric_hmc

    check_dense_metric_hmc(FakeOps(), C=8, D=6)


def test_tempered_smc_driver():
    from tests.sampler_parity import check_smc_binomial

    check_smc_binomial(FakeOps(), 400, 8, bk.metropolis_kernel(0.5), mean_atol=0.03, var_atol=0.004)
    check_smc_binomial(FakeOps(), 300, 6, bk.mala_kernel(0.15, 2), mean_atol=0.04, var_atol=0.005)
    check_smc_binomial(FakeOps(), 300, 6, bk.hmc_kernel(0.4, 3), mean_atol=0.04, var_atol=0.005)


def test_logistic_target_driver():
    from tests.sampler_parity import check_logistic_target

    check_logistic_target(FakeOps(), N=120, D=6, C=12)


def test_batched_model_output_is_validated():
    import torch

    class F32:
        batched = True

        def dims(self):
            return 4

        def log_density(self, Th):
            return -0.5 * (Th * Th).sum(dim=1)

        def log_density_gradient(self, Th):
            return (-0.5 * (Th * Th).sum(dim=1)).reshape(-1, 1), (-Th).to(torch.float32)  # f32 grad, (C,1) logp

    ops = FakeOps()
    a = bk.HMCDiag(F32(), 0.1, 3, chains=5, seed=1, ops=ops)
    b = bk.HMCDiag(bk.IsoGaussian(4, ops=ops), 0.1, 3, chains=5, seed=1, path="step", ops=ops)
    ta, _ = a.sample()
    tb, _ = b.sample()
    np.testing.assert_allclose(ta.numpy(), tb.numpy(), rtol=1e-6)  # gradient rounded through f32

    class Bad(F32):
        def log_density_gradient(self, Th):
            return -0.5 * (Th * Th).sum(dim=1), -Th[:, :2]

    with pytest.raises(ValueError):
        bk.HMCDiag(Bad(), 0.1, 3, chains=5, seed=1, ops=ops).sample()


# ---- Metropolis / Metropolis-Hastings drop-ins: the behaviours test/test_metropolis.py pins -------
def _rw(seed, scale=4.0):
    g = np.random.default_rng(seed=seed)
    return lambda theta: g.normal(loc=theta, scale=scale)


def test_metropolis_reproducible_and_seed_sensitive():
    # test_metropolis.py:171-196
    from oracle.models import StdNormal

    init = np.array([0.3])
    runs = []
    for seed in (1848, 1848, 1912):
        s = bk.Metropolis(StdNormal(), proposal_fn=_rw(12345), init=init, seed=seed, ops=FakeOps())
        runs.append(np.array([s.sample()[0] for _ in range(25)]))
    np.testing.assert_array_equal(runs[0], runs[1])
    assert not np.array_equal(runs[0], runs[2])


def test_metropolis_hastings_protocol():
    # test_metropolis.py:257-295: iter returns self, next() == sample(), bad proposals raise ValueError
    from oracle.models import StdNormal

    mh = bk.MetropolisHastings(StdNormal(), lambda x: 1, lambda x, y: 1, ops=FakeOps())
    assert iter(mh) is mh
    init = np.array([-0.7])
    a = bk.MetropolisHastings(StdNormal(), _rw(123), lambda o, g: 1, init=init, seed=996, ops=FakeOps())
    b = bk.MetropolisHastings(StdNormal(), _rw(123), lambda o, g: 1, init=init, seed=996, ops=FakeOps())
    np.testing.assert_array_equal(np.array([a.sample()[0] for _ in range(25)]),
                                  np.array([next(b)[0] for _ in range(25)]))
    bad = bk.MetropolisHastings(StdNormal(), lambda x: "a", lambda x, y: 1, ops=FakeOps())
    with pytest.raises(ValueError):
        bad.sample()


def test_metropolis_hastings_with_symmetric_proposal_equals_metropolis():
    # test_equivalencies.py:35-60
    from oracle.models import StdNormal

    init = np.array([0.1])
    m = bk.Metropolis(StdNormal(), _rw(5, 1.0), init=init, seed=99, ops=FakeOps())
    mh = bk.MetropolisHastings(StdNormal(), _rw(5, 1.0), lambda o, g: -0.5 * float((o - g) @ (o - g)),
                               init=init, seed=99, ops=FakeOps())
    np.testing.assert_array_equal(np.array([m.sample()[0] for _ in range(50)]),
                                  np.array([mh.sample()[0] for _ in range(50)]))


def test_accept_test_functions_many_chains():
    # metropolis.py:12-76 on (C,) tensors: strict `<` against log(u) of every chain's own stream
    import torch

    from bayes_kit_amd.metropolis import ChainRng, metropolis_accept_test, metropolis_hastings_accept_test

    C = 9
    ops = FakeOps()
    rng = ChainRng(31, C, ops=ops)
    logu = np.array([np.log(np.random.Generator(np.random.Philox(key=[31, c])).uniform()) for c in range(C)])
    lp_cur = torch.zeros(C, dtype=torch.float64)
    # just above / exactly at / just below the boundary
    delta = np.where(np.arange(C) % 3 == 0, np.nextafter(logu, 0.0), np.where(np.arange(C) % 3 == 1, logu,
                                                                                  np.nextafter(logu, -np.inf)))
    got = metropolis_accept_test(torch.from_numpy(delta), lp_cur, rng)
    assert got.tolist() == [bool(l < d) for l, d in zip(logu, delta)]
    assert got.tolist() == [i % 3 == 0 for i in range(C)]
    # second uniform of each stream, with transition terms
    g2 = [np.random.Generator(np.random.Philox(key=[31, c])) for c in range(C)]
    for g in g2:
        g.uniform()
    logu2 = np.array([np.log(g.uniform()) for g in g2])
    fwd, rev = np.linspace(-1, 1, C), np.linspace(0.5, -2, C)
    lp_p = np.linspace(-3, 0.2, C)
    got = metropolis_hastings_accept_test(torch.from_numpy(lp_p), lp_cur, torch.from_numpy(fwd),
                                          torch.from_numpy(rev), rng)
    assert got.tolist() == [bool(l < (p - 0.0) + (r - f)) for l, p, f, r in zip(logu2, lp_p, fwd, rev)]


# ---- drop-in behaviours pinned by the reference's own tests (shared bodies, fake ops here) ----------
def test_reference_test_behaviours_with_fake_ops():
    from tests import dropin_behaviours as db

    ops = FakeOps()
    db.check_end_pos_pairs(ops)
    db.check_accept_tests_with_host_rng(ops)
    db.check_theta_initialization(ops)
    db.check_smc_with_reference_style_model(ops)


def test_checkpoint_of_sampler_moments_recorder_and_draw_store(tmp_path):
    from tests.diag_parity import check_checkpoint_of_sampler_and_diagnostics

    check_checkpoint_of_sampler_and_diagnostics(FakeOps(), str(tmp_path), chains=6, D=5, draws=24, at=11)


@pytest.mark.parametrize("K", [2, 3, 4])
def test_drghmc_device_side_lists_equal_host_sized_launches(K):
    """Lane lists built by the appending entry points (bk_dr_accept_prob_test_next / _ghost_next: here in
    REVERSE lane order, on the device in wavefront-timing order) against stable compaction sized by host
    reads: same draws, momenta, stream positions and lane sets -- lane order carries no meaning.  K = 4
    exercises the alternating list buffers of a level (ghost lists two deep)."""
    sizes, counts = [0.5, 0.2, 0.08, 0.03][:K], [2, 3, 5, 7][:K]
    ops_a, ops_b = FakeOps(), FakeOps()
    a = bk.DrGhmcDiag(bk.Funnel(7, ops=ops_a), K, sizes, counts, 0.4, chains=40, seed=11, device_counts=False, ops=ops_a)
    b = bk.DrGhmcDiag(bk.Funnel(7, ops=ops_b), K, sizes, counts, 0.4, chains=40, seed=11, device_counts=True, ops=ops_b)
    ops_u = FakeOps()
    u = bk.DrGhmcDiag(bk.Funnel(7, ops=ops_u), K, sizes, counts, 0.4, chains=40, seed=11, device_counts=True,
                      fuse_first_ghost=False, recompute_gradient=False, ops=ops_u)   # every ghost a launch of its own, cached gradients
    # model-opaque: the gradient a separate, COUNTED op per leapfrog step (drghmc.py:280-283), lane counts on the device
    ops_o = FakeOps()
    o = bk.DrGhmcDiag(bk.Funnel(7, ops=ops_o), K, sizes, counts, 0.4, chains=40, seed=11, device_counts=True,
                      path="step", ops=ops_o)
    assert b._dev_counts and not a._dev_counts and b._one_launch
    assert o._dev_counts and not o._one_launch and not o._fused and o.host_syncs_per_draw == 0
    seen = set()
    for n in range(10):
        ta, la = a.sample()
        tb, lb = b.sample()
        tu, lu = u.sample()
        to, lo = o.sample()
        assert np.array_equal(ta.numpy(), tb.numpy()) and np.array_equal(la.numpy(), lb.numpy()), (K, n)
        assert np.array_equal(ta.numpy(), tu.numpy()) and np.array_equal(la.numpy(), lu.numpy()), (K, n)
        assert np.array_equal(ta.numpy(), to.numpy()) and np.array_equal(la.numpy(), lo.numpy()), (K, n)
        assert a.last_stage_lanes == b.last_stage_lanes == u.last_stage_lanes == o.last_stage_lanes
        assert a.last_lane_steps == b.last_lane_steps == o.last_lane_steps and a.last_grad_evals == o.last_grad_evals
        seen.update(t for t, _ in a.last_stage_lanes)
    assert np.array_equal(a._rho.numpy(), b._rho.numpy()) and np.array_equal(a._rho.numpy(), o._rho.numpy())
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    np.testing.assert_array_equal(a.rng_state(), o.rng_state())
    assert len(seen) >= min(4, 2 ** (K - 1))
    # the counted draw makes exactly the reference's model calls: one gradient op per leapfrog step of every trajectory
    # in the schedule (empty lane sets included: the launch sequence is fixed), and never reads a count back
    assert ops_o.calls["target_grad"] == 1 + 10 * sum(st for _, st in o._schedule)


def test_drghmc_counted_steps_on_a_gaussian_with_metric_equal_host_sized_launches():
    """The same on DiagGaussian with a diagonal metric and no probabilistic retry: the counted step-by-step draw (gradient a
    separate op, and {gradient, kick, drift} as one launch per step) and the one-launch proposals (round 5: a separable density
    is a lanes-form density without head coordinates) against the host-sized, stably compacted draw."""
    lam = np.linspace(0.5, 3.0, 9)
    met = np.linspace(0.8, 1.3, 9)
    mk = lambda ops, dc, **kw: bk.DrGhmcDiag(bk.DiagGaussian(lam, ops=ops), 3, [0.9, 0.4, 0.15], [2, 4, 6], 0.5, metric_diag=met,  # noqa: E731
                                             chains=33, seed=3, prob_retry=False, device_counts=dc, ops=ops, **kw)
    a = mk(FakeOps(), False, path="opaque")
    o = mk(FakeOps(), True, path="opaque")
    h = mk(FakeOps(), True, path="step")
    f = mk(FakeOps(), True)
    assert o._dev_counts and not o._one_launch and not a._dev_counts and h._step_hook and not o._step_hook and f._one_launch
    for n in range(12):
        ta, la = a.sample()
        to, lo = o.sample()
        assert np.array_equal(ta.numpy(), to.numpy()) and np.array_equal(la.numpy(), lo.numpy()), n
        assert a.last_stage_lanes == o.last_stage_lanes
        for s_ in (h, f):
            ts, ls = s_.sample()
            assert np.array_equal(ta.numpy(), ts.numpy()) and np.array_equal(la.numpy(), ls.numpy()), n
            assert a.last_stage_lanes == s_.last_stage_lanes
    assert np.array_equal(a._rho.numpy(), o._rho.numpy())
    np.testing.assert_array_equal(a.rng_state(), o.rng_state())


def test_drghmc_attached_diagnostics_equal_manual_updates():
    from tests.sampler_parity import check_attached_diagnostics

    check_attached_diagnostics(FakeOps())                         # device-side lists (eager on the CPU stand-in)
    check_attached_diagnostics(FakeOps(), device_counts=False)    # host-sized path: fed after the draw


def test_recorder_dims_square_draws_padded_moments_and_attach_after_restore():
    from tests.sampler_parity import check_recorder_and_moments_edges

    check_recorder_and_moments_edges(FakeOps())


@pytest.mark.parametrize("source", ["np.random", "replay"])
@pytest.mark.parametrize("name", ["smc_ref_binomial", "smc_gauss5_m512"])
def test_smc_reference_stream_host_logic(name, source):
    # the driver's order of stream consumption and its arithmetic, on the CPU stand-in for the kernels
    from tests.sampler_parity import check_smc_reference_stream

    check_smc_reference_stream(FakeOps(), name, source)


def test_smc_reference_stream_refuses_what_the_reference_cannot_do():
    init = np.zeros((8, 2))
    model = bk.TorchPriorLikelihoodModel(lambda T: -(T * T).sum(1), lambda T: -(T * T).sum(1), 2)
    with pytest.raises(ValueError, match="metropolis_kernel"):
        bk.TemperedLikelihoodSMC(model, 8, 3, init, bk.mala_kernel(0.1), seed=np.random, ops=FakeOps())
    with pytest.raises(ValueError, match="one rank"):
        bk.TemperedLikelihoodSMC(model, 8, 3, init, bk.metropolis_kernel(0.1), seed=np.random.RandomState(1), slot_id0=8,
                                 ops=FakeOps())


def test_smc_reweighting_with_no_weight_left_raises_like_numpy_choice():
    import torch

    model = bk.TorchPriorLikelihoodModel(lambda T: -(T * T).sum(1), lambda T: torch.full((T.shape[0],), float("-inf"), dtype=torch.float64), 1)
    smc = bk.TemperedLikelihoodSMC(model, 6, 3, np.zeros((6, 1)), bk.metropolis_kernel(0.1), seed=np.random.RandomState(3), ops=FakeOps())
    with pytest.raises(FloatingPointError, match="weights sum to"):
        smc.transition(1)


def test_adaptive_ladder_accepts_zero_weight_particles():
    # a -inf log likelihood is a hard constraint (weight 0), not an error; NaN / +inf are
    import torch

    from bayes_kit_amd.smc import TemperedLikelihoodSMC

    model = bk.TorchPriorLikelihoodModel(lambda T: -(T * T).sum(1), lambda T: -(T * T).sum(1), 1)
    smc = TemperedLikelihoodSMC(model, 6, 3, np.zeros((6, 1)), bk.metropolis_kernel(0.1), seed=1, adaptive=0.5, ops=FakeOps())
    ll = torch.tensor([-1.0, -2.0, float("-inf"), -0.5, -3.0, float("-inf")], dtype=torch.float64)
    d = smc._next_temperature(ll)
    assert 0.0 < d <= 1.0
    for bad in (float("nan"), float("inf")):
        ll2 = ll.clone()
        ll2[0] = bad
        with pytest.raises(FloatingPointError):
            smc._next_temperature(ll2)
    with pytest.raises(FloatingPointError):
        smc._next_temperature(torch.full((6,), float("-inf"), dtype=torch.float64))


def test_adaptive_smc_ladder_keeps_its_ess_and_finds_the_posterior():
    from tests.sampler_parity import check_adaptive_smc_ladder

    check_adaptive_smc_ladder(FakeOps())


def test_logistic_retemper_equals_a_fresh_evaluation():
    from tests.sampler_parity import check_logistic_retemper

    check_logistic_retemper(FakeOps())


def test_torch_model_with_a_written_out_gradient_equals_autograd():
    """TorchModel(fn, D, grad_fn=...): no autograd graph, and a leapfrog step calls grad_fn alone (the engine's
    gradient-only hook); for the diagonal Gaussian the written-out gradient -(lam * theta) is autograd's bit for bit."""
    import torch

    lam = torch.logspace(0, 1, 9, dtype=torch.float64)
    calls = {"fn": 0, "grad": 0}

    def fn(Th):
        calls["fn"] += 1
        return -0.5 * (Th * Th * lam).sum(dim=1)

    def grad_fn(Th):
        calls["grad"] += 1
        return -(lam * Th)

    for mk in (lambda m: bk.HMCDiag(m, 0.2, 5, chains=17, seed=3, ops=FakeOps()),
               lambda m: bk.MALA(m, 0.05, chains=17, seed=3, ops=FakeOps()),
               lambda m: bk.DrGhmcDiag(m, 2, [0.4, 0.15], [2, 4], 0.5, chains=17, seed=3, ops=FakeOps())):
        a = mk(bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), 9))
        b = mk(bk.TorchModel(fn, 9, grad_fn=grad_fn))
        for n in range(6):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (type(a).__name__, n)
    assert calls["grad"] > calls["fn"] > 0   # trajectories ask for the gradient alone
    # layout="dc": both functions take the (D, C) array
    a = bk.HMCDiag(bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), 9, layout="dc"), 0.2, 5, chains=8, seed=4,
                   ops=FakeOps())
    b = bk.HMCDiag(bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam[:, None]).sum(dim=0), 9, layout="dc",
                                 grad_fn=lambda Th: -(lam[:, None] * Th)), 0.2, 5, chains=8, seed=4, ops=FakeOps())
    for n in range(4):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb), n
