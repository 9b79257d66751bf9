"""Provider x sampler parity bodies (tests/provider_parity.py) on the CPU stand-in ops: the host-side
bridge between user models (autograd, own gradient layouts) and the samplers, small sizes.  The same
bodies run through the HIP library in tests/test_gpu_providers.py."""
import pytest

from tests import provider_parity as pp
from tests.fake_ops import FakeOps


@pytest.fixture()
def ops():
    return FakeOps()


def test_torch_autograd_diag_gaussian_under_mala_and_drghmc(ops):
    stages = pp.check_torch_diag_gaussian(ops, C=13, D=5, draws=6)
    assert "P0" in stages


def test_torch_autograd_funnel_under_all_samplers(ops):
    pp.check_torch_funnel(ops, C=9, D=5, draws=5, cfg4_steps=False)


def test_torch_autograd_logistic_under_all_samplers(ops):
    pp.check_torch_logistic(ops, N=40, D=4, C=7, draws=4)


def test_user_gradient_layouts_under_all_samplers(ops):
    pp.check_gradient_layouts(ops, C=11, D=6, draws=5)
