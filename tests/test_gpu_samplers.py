"""Parity of the HIP samplers with the reference's golden vectors and with the oracle."""
import numpy as np
import pytest
import torch

import bayes_kit_amd as bk
from tests.sampler_parity import check_many_chain, check_single_chain_host_model

pytestmark = pytest.mark.gpu

MANY = ["hmc_stdnormal", "hmc_steps0", "hmc_iso4", "hmc_iso128_cfg2", "hmc_diag16_metric", "hmc_diag1024_cfg3",
        "mala_stdnormal", "mala_iso8", "mala_diag16", "mala_init"]


@pytest.fixture(scope="module")
def ops():
    return bk._lib.default_ops()


@pytest.mark.parametrize("name", MANY)
def test_many_chain_vs_reference_golden(name, ops):
    check_many_chain(name, ops)


@pytest.mark.parametrize("name", ["hmc_pcg_seed", "hmc_iso4", "mala_stdnormal", "mala_init"])
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
