"""Behaviours the reference's own tests pin that are not sampler trajectories: written in this
repo's words, run with the fake ops on CPU (tests/test_host_logic.py) and with the HIP library
on the GPU (tests/test_gpu_samplers.py).  Each cites the reference test it mirrors."""
from unittest.mock import Mock

import numpy as np
import torch

import bayes_kit_amd as bk
from bayes_kit_amd.iat import _end_pos_pairs
from bayes_kit_amd.metropolis import metropolis_accept_test, metropolis_hastings_accept_test


def _with_default_ops(ops, fn):
    """Entry points without an ops= argument use the process-wide ops object."""
    old = bk._lib._default_ops
    bk._lib._default_ops = ops
    try:
        return fn()
    finally:
        bk._lib._default_ops = old


def check_end_pos_pairs(ops):
    # test/test_iat.py:72-80 (and the examples of iat.py:19-28)
    for acor, want in [([], 0), ([1], 0), ([1, 0.4], 2), ([1, -0.5], 2), ([1, -0.5, 0.25], 2),
                       ([1, -0.5, 0.25, -0.3], 2), ([1, -0.5, 0.25, -0.1], 4),
                       ([1, -0.5, 0.25, -0.3, 0.05], 2), ([1, -0.5, 0.25, -0.1, 0.05], 4)]:
        assert _end_pos_pairs(acor, ops=ops) == want, acor
    # many chains at once: one column per chain
    cols = np.array([[1, 1, 1], [-0.5, -0.5, 0.4], [0.25, 0.25, 0.3], [-0.3, -0.1, 0.2], [0.05, 0.05, -0.9]])
    got = _end_pos_pairs(torch.from_numpy(cols).to(ops.device), ops=ops)
    assert got.cpu().tolist() == [2, 4, 4]


def check_accept_tests_with_host_rng(ops):
    # test/test_metropolis.py:19-103: a host rng (here: mocks returning a fixed uniform) and floats
    def body():
        top = Mock()
        top.uniform = Mock(return_value=1)  # log(1) = 0: still accepted whenever lp_prop > lp_cur
        assert metropolis_accept_test(-0.2, -0.7, top) is True
        low = Mock()
        low.uniform = Mock(return_value=0.5)
        assert metropolis_accept_test(np.log(0.4), np.log(0.81), low) is False   # ratio 0.49 < 0.5
        assert metropolis_accept_test(np.log(0.41), np.log(0.81), low) is True   # ratio 0.506 > 0.5
        bal = np.log(0.5)
        assert not metropolis_hastings_accept_test(np.log(0.4), np.log(0.81), bal, bal, low)
        assert metropolis_hastings_accept_test(np.log(0.4), np.log(0.81), np.log(0.3), np.log(0.6), low)
        # equal transition terms reduce to the plain rule (test_metropolis.py:88-103)
        g1, g2 = np.random.default_rng(3), np.random.default_rng(3)
        for lp_p, lp_c in [(-1.0, -1.2), (-2.0, -1.0), (-0.3, -0.31), (-5.0, -0.1)]:
            assert metropolis_accept_test(lp_p, lp_c, g1) == metropolis_hastings_accept_test(lp_p, lp_c, -0.7, -0.7, g2)
        assert g1.bit_generator.state == g2.bit_generator.state  # exactly one uniform each per call

    _with_default_ops(ops, body)


def check_theta_initialization(ops):
    # test/test_theta_initialization.py:17-54: only dims() matters; Mock models (whose every
    # attribute is truthy, and whose log_density returns a Mock) must construct
    def make(init, dims=1):
        m = Mock()
        m.dims = Mock(return_value=dims)
        m.log_density_gradient = Mock(return_value=(0.5, (0,)))
        return [bk.HMCDiag(m, stepsize=0.25, steps=10, init=init, ops=ops),
                bk.MALA(m, epsilon=0.5, init=init, ops=ops),
                bk.Metropolis(m, lambda x: 1, init=init, ops=ops),
                bk.MetropolisHastings(m, lambda x: 1, lambda x, y: 1, init=init, ops=ops)]

    for s in make(np.array([])):  # an empty init is no init
        assert np.asarray(s._theta).shape == (1,)
    for s in make(np.array([3])):
        np.testing.assert_array_equal(s._theta, np.array([3]))
    for s in make(np.array([3, 3, 3]), dims=3):
        np.testing.assert_array_equal(s._theta, np.array([3, 3, 3]))


def check_smc_with_reference_style_model(ops, M=75, N=15):
    # test/test_tempered_smc.py:8-30: per-particle host model, callable initial state,
    # metropolis_kernel; thetas come back as a NumPy array.  Loose moment check (M is small).
    from tests.host_models import Binomial

    model = Binomial(alpha=2, beta=3, x=5, N=15)
    gen = np.random.default_rng(5)
    smc = bk.TemperedLikelihoodSMC(model, M, N, lambda i: gen.normal(size=1), bk.metropolis_kernel(0.5), seed=17,
                                   ops=ops)
    smc.run()
    th = smc.thetas
    assert isinstance(th, np.ndarray) and th.shape == (M, 1)
    p = 1.0 / (1.0 + np.exp(-th[:, 0]))
    post_mean = (2 + 5) / (2 + 3 + 15)
    assert abs(p.mean() - post_mean) < 0.08, p.mean()
