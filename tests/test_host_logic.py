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
        "mala_diag16", "mala_init"]
SINGLE = ["hmc_pcg_seed", "hmc_iso4", "mala_stdnormal", "mala_init"]


@pytest.mark.parametrize("name", MANY)
def test_many_chain_driver_vs_golden(name):
    check_many_chain(name, FakeOps())


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
    ops = FakeOps()
    a = bk.HMCDiag(tm, 0.05, 8, chains=6, seed=9, ops=ops)
    b = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 8, chains=6, seed=9, ops=ops)
    for _ in range(10):
        ta, la = a.sample()
        tb, lb = b.sample()
        np.testing.assert_allclose(ta.numpy(), tb.numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(la.numpy(), lb.numpy(), rtol=1e-11)


def test_no_gpu_no_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle.models import StdNormal

    with pytest.raises(bk._lib.BkHipError):
        bk.HMCDiag(StdNormal(), 0.1, 3)
