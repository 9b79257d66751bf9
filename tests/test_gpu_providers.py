"""User-model providers x samplers on the GPU: the Model protocol is consumed by all
three samplers (bayes_kit/hmc.py:45-50, mala.py:31-32,46-48, drghmc.py:243-247,280-288), so every provider
-- PyTorch autograd incl. transcendental densities, user gradient layouts, the compiled plugin -- is run
under MALA and DRGHMC (and HMC) against the oracle samplers driving the NumPy twin of the model."""
import numpy as np
import pytest
import torch

import bayes_kit_amd as bk
from tests import provider_parity as pp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    return bk._lib.default_ops()


def test_torch_autograd_diag_gaussian_under_mala_and_drghmc(ops):
    stages = pp.check_torch_diag_gaussian(ops)
    assert {"P0", "P1", "G0(P1)"} <= stages, stages


def test_torch_autograd_funnel_under_all_samplers(ops):
    stages = pp.check_torch_funnel(ops)
    assert "P0" in stages


def test_torch_autograd_logistic_under_all_samplers(ops):
    pp.check_torch_logistic(ops)


def test_user_gradient_layouts_under_all_samplers(ops):
    pp.check_gradient_layouts(ops)


def test_compiled_plugin_under_mala_and_drghmc(ops):
    stages = pp.check_plugin_target(ops)
    assert {"P0", "P1", "G0(P1)"} <= stages, stages


def test_mala_two_pass_is_the_path_user_models_take_at_scale(ops):
    """A user autograd model with Philox streams and D >= 32 takes the two-pass MALA draw (bk_mala_step reads
    the relaid-out gradient); same draws as the step-by-step kernels and as the built-in target."""
    D, C = 64, 512
    lam = np.logspace(0, 1, D)
    a = bk.MALA(pp.torch_diag_gaussian(lam, ops.device), 0.004, chains=C, seed=5)
    b = bk.MALA(pp.torch_diag_gaussian(lam, ops.device), 0.004, chains=C, seed=5, two_pass=False)
    c = bk.MALA(bk.DiagGaussian(lam), 0.004, chains=C, seed=5)
    r = bk.MALA(pp.RowMajorDiag(lam, ops.device), 0.004, chains=C, seed=5)
    assert a.path.startswith("two-pass") and b.path == "step-by-step" and r.path.startswith("two-pass")
    for n in range(6):
        ta, la = a.sample()
        for s in (b, c, r):
            t, l = s.sample()
            assert torch.equal(ta, t), n
            np.testing.assert_allclose(la.cpu().numpy(), l.cpu().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(a.rng_state(), c.rng_state())
