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


DIAG_SRC = """
__device__ __forceinline__ void bk_term(double th, i64 d, const double* lam, double& term, double& grad) {
  const double t = lam[d] * th;
  term = -0.5 * (th * t);
  grad = -t;
}
"""

FUNNEL_SRC = """
// Neal's funnel, one chain per call, coordinates summed in order (bk.Funnel's order past 128 coordinates)
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* /*params*/) {
  const double v = th[0];
  double s = 0.0;
  for (i64 d = 1; d < D; ++d) { const double x = th[d]; s = s + x * x; }
  const double ev = exp(-v), hn = 0.5 * (double)(D - 1), he = 0.5 * ev;
  if (g.wanted()) {
    g.set(0, ((-v / 9.0) - hn) + he * s);
    for (i64 d = 1; d < D; ++d) g.set(d, -(ev * th[d]));
  }
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}
"""


def test_density_compiled_from_source_under_every_sampler(ops):
    """CTarget.from_source: a density written as a few lines of HIP C++, compiled with hipcc at
    construction into the plugin ABI (both forms: host-sized and counted).  The elementwise form of the diagonal
    Gaussian is the built-in target bit for bit (same operation order, the library's own order for the per-chain
    sum) under HMC, MALA and DRGHMC -- incl. device-side lane counts inside one hipGraph; the per-chain form of
    Neal's funnel is bk.Funnel's sequential-sum variant (D > 129) bit for bit."""
    import torch

    D, C = 48, 1500
    lam = np.logspace(0, 2, D)
    lam_d = torch.from_numpy(lam).to(ops.device)
    src = lambda: bk.CTarget.from_source(DIAG_SRC, D, params=lam_d)  # noqa: E731
    assert src().bk_counted
    pairs = [
        (bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 9, chains=C, seed=3, fuse_builtin=False), bk.HMCDiag(src(), 0.02, 9, chains=C, seed=3)),
        (bk.MALA(bk.DiagGaussian(lam), 2e-3, chains=C, seed=4), bk.MALA(src(), 2e-3, chains=C, seed=4)),
        (bk.DrGhmcDiag(bk.DiagGaussian(lam), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5),
         bk.DrGhmcDiag(src(), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5)),
    ]
    assert pairs[2][1]._dev_counts and pairs[2][1]._use_graph and not pairs[2][1]._one_launch
    for a, b in pairs:
        for n in range(8):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (type(a).__name__, n)
        np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    # odd chain counts and unaligned views take the one-chain-per-lane kernels
    th = torch.randn((D, 333), dtype=torch.float64, device=ops.device)
    g1, g2 = torch.empty_like(th), torch.empty_like(th)
    l1, l2 = torch.empty(333, dtype=torch.float64, device=ops.device), torch.empty(333, dtype=torch.float64, device=ops.device)
    bk.DiagGaussian(lam).bk_eval(th, g1, l1)
    src().bk_eval(th, g2, l2)
    assert torch.equal(g1, g2) and torch.equal(l1, l2)
    src().bk_eval(th, g2.zero_(), None)
    assert torch.equal(g1, g2)
    # the per-chain form
    Df = 150
    fs = bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain")
    a = bk.DrGhmcDiag(bk.Funnel(Df), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9, fuse_builtin=False, device_counts=False)
    b = bk.DrGhmcDiag(fs, 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9)
    assert b._dev_counts and b._use_graph
    for n in range(8):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb), ("funnel from source", n)
    # a source that does not compile says so (hipcc's message), it does not fall back to anything
    with pytest.raises(bk._lib.BkHipError):
        bk.CTarget.from_source("this is not C++", 3)
