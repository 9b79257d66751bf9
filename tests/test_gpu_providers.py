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
// Neal's funnel, one chain per call, coordinates summed in the library's canonical class order (bk.Funnel's order:
// 16 interleaved class sums -> 4 group sums -> total)
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* /*params*/) {
  const double v = th[0];
  double cs[16];
  for (int c = 0; c < 16; ++c) {
    double a = 0.0;
    for (i64 d = 1 + c; d < D; d += 16) { const double x = th[d]; a = a + x * x; }
    cs[c] = a;
  }
  double q[4];
  for (int k = 0; k < 4; ++k) q[k] = ((cs[k] + cs[k + 4]) + cs[k + 8]) + cs[k + 12];
  const double s = ((q[0] + q[1]) + q[2]) + q[3];
  const double ev = bk_exp(-v), hn = 0.5 * (double)(D - 1), he = 0.5 * ev;
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
    Neal's funnel, summed in the library's class order, is bk.Funnel bit for bit."""
    import torch

    D, C = 48, 1500
    lam = np.logspace(0, 2, D)
    lam_d = torch.from_numpy(lam).to(ops.device)
    src = lambda: bk.CTarget.from_source(DIAG_SRC, D, params=lam_d)  # noqa: E731
    assert src().bk_counted
    pairs = [
        (bk.HMCDiag(bk.DiagGaussian(lam), 0.02, 9, chains=C, seed=3, path="step"), bk.HMCDiag(src(), 0.02, 9, chains=C, seed=3)),
        (bk.MALA(bk.DiagGaussian(lam), 2e-3, chains=C, seed=4), bk.MALA(src(), 2e-3, chains=C, seed=4)),
        (bk.DrGhmcDiag(bk.DiagGaussian(lam), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5),
         bk.DrGhmcDiag(src(), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5)),
    ]
    # (round 5) a separable density is also a lane-spread one without head coordinates: DrGhmcDiag runs its proposals as ONE
    # launch each (D <= 128); with path="step" it steps, one launch per leapfrog step; path="opaque": gradient op per step
    pairs.append((pairs[2][0], bk.DrGhmcDiag(src(), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5, path="step")))
    pairs.append((pairs[2][0], bk.DrGhmcDiag(src(), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5, path="opaque")))
    assert pairs[2][1]._dev_counts and pairs[2][1]._use_graph and pairs[2][1]._one_launch and pairs[2][1].host_syncs_per_draw == 0
    assert pairs[3][1]._step_hook and not pairs[3][1]._one_launch and not pairs[4][1]._step_hook
    for i, (a, b) in enumerate(pairs):
        if i >= 3:   # (pairs 3, 4 share their reference sampler with pair 2: a fresh one, stepping with the gradient op)
            a = bk.DrGhmcDiag(bk.DiagGaussian(lam), 3, [0.2, 0.08, 0.03], [3, 6, 12], 0.3, chains=C, seed=5, path="opaque")
        for n in range(8):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb), (i, type(a).__name__, n)
            if i == 2:  # (the one-launch kernel sums the log density and the kinetic energy in its lanes' order)
                torch.testing.assert_close(la, lb, rtol=1e-12, atol=1e-12)
            else:
                assert torch.equal(la, lb), (i, type(a).__name__, n)
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
    # the per-chain form: a chain's coordinates staged in its lane's registers (D <= 128), in LDS (D <= 300), or read from
    # global memory as the user's loops ask for them (larger D) -- the same values
    for Df in (150, 101, 17, 128, 350):
        fs = bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain")
        a = bk.DrGhmcDiag(bk.Funnel(Df), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9, path="step", device_counts=False)
        b = bk.DrGhmcDiag(fs, 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9)
        b2 = bk.DrGhmcDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9,
                           path="opaque", metric_diag=None)
        hm = bk.HMCDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 0.05, 5, chains=333, seed=4,
                        metric_diag=np.linspace(0.8, 1.3, Df))
        hs = bk.HMCDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 0.05, 5, chains=333, seed=4, path="opaque",
                        metric_diag=np.linspace(0.8, 1.3, Df))
        # (D <= 128) the whole trajectory of a proposal is one launch (bk_leapfrog_trajectory: theta in registers, rho in LDS);
        # path="step": one launch per leapfrog step; path="opaque": gradient op + kick+drift per step
        b3 = bk.DrGhmcDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9,
                           path="step")
        b4 = bk.DrGhmcDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9,
                           device_counts=False)
        h3 = bk.HMCDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 0.05, 5, chains=333, seed=4, path="step",
                        metric_diag=np.linspace(0.8, 1.3, Df))
        assert b._dev_counts and b._use_graph and b._step_hook == (Df <= 128) and not b2._step_hook and hm._step_hook == (Df <= 128)
        assert b._traj_hook == (Df <= 128) and hm._traj_hook == (Df <= 128) and b4._traj_hook == (Df <= 128) and not b4._dev_counts
        assert not b3._traj_hook and not h3._traj_hook and h3._step_hook == (Df <= 128)
        for n in range(8):
            ta, la = a.sample()
            tb, lb = b.sample()
            tb2, _ = b2.sample()
            tb3, _ = b3.sample()
            tb4, _ = b4.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb) and torch.equal(ta, tb2), ("funnel from source", Df, n)
            assert torch.equal(ta, tb3) and torch.equal(ta, tb4), ("funnel from source, step / host-sized trajectory", Df, n)
            t1, l1 = hm.sample()
            t2, l2 = hs.sample()
            t3, l3 = h3.sample()
            assert torch.equal(t1, t2) and torch.equal(l1, l2), ("HMC, per-chain source, one launch per trajectory", Df, n)
            assert torch.equal(t1, t3) and torch.equal(l1, l3), ("HMC, per-chain source, one launch per step", Df, n)
        assert b._grad_calls == b3._grad_calls == b2._grad_calls and hm._grad_calls == hs._grad_calls == h3._grad_calls
        # a trajectory of ONE step (no in-kernel step loop: the gathering first step, then the last gradient) and of two
        for L in (1, 2):
            k1 = bk.HMCDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 0.05, L, chains=333, seed=14)
            k2 = bk.HMCDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain"), 0.05, L, chains=333, seed=14, path="opaque")
            for n in range(3):
                t1, l1 = k1.sample()
                t2, l2 = k2.sample()
                assert torch.equal(t1, t2) and torch.equal(l1, l2), ("HMC, per-chain source", Df, L, n)
        if Df <= 128:
            # stage="lds": the coordinates staged in LDS in every kernel, the one-launch step and trajectory kernels included (for
            # long functions; measured slower for this short one) -- the same draws
            mk_l = lambda **kw: bk.DrGhmcDiag(bk.CTarget.from_source(FUNNEL_SRC, Df, form="chain", stage="lds"), 2, [0.3, 0.1], [3, 6],  # noqa: E731
                                              0.3, chains=700, seed=9, **kw)
            c0 = bk.DrGhmcDiag(bk.Funnel(Df), 2, [0.3, 0.1], [3, 6], 0.3, chains=700, seed=9, path="step", device_counts=False)
            l1, l2, l3 = mk_l(), mk_l(path="step"), mk_l(path="opaque")
            assert l1._traj_hook and l2._step_hook and not l2._traj_hook and not l3._step_hook
            for n in range(5):
                t0, p0 = c0.sample()
                for o in (l1, l2, l3):
                    to, po = o.sample()
                    assert torch.equal(t0, to) and torch.equal(p0, po), ("stage=lds", Df, n)
        # (a call with another D than the one compiled for takes the unstaged path)
        th = torch.randn((Df - 1, 130), dtype=torch.float64, device=ops.device)
        g1, g2 = torch.empty_like(th), torch.empty_like(th)
        bk.Funnel(Df - 1).bk_eval(th, g1, None)
        bk.CTarget(fs.source_library, "bk_src_target", Df - 1).bk_eval(th, g2, None)
        assert torch.equal(g1, g2), Df
    # a source that does not compile says so (hipcc's message), it does not fall back to anything
    with pytest.raises(bk._lib.BkHipError):
        bk.CTarget.from_source("this is not C++", 3)


# ---- round 5: the library's one-launch kernels for ANY compiled-source density --------------------------------------
FUNNEL_LANES_SRC = """
// Neal's funnel for the lane-spread form (head = 1: v = theta_0 is held by every lane of the chain)
template <class L>
__device__ double bk_lanes_density(L& c, const double* /*params*/) {
  const double v = c.head(0);
  const double s = c.sum([](double x, i64) { return x * x; });
  const double ev = bk_exp(-v);
  const double hn = 0.5 * (double)(c.dims() - 1);
  const double he = 0.5 * ev;
  c.grad_head(0, ((-v / 9.0) - hn) + he * s);
  c.grad([ev](double x, i64) { return -(ev * x); });
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}
"""

# a hierarchical model that is NOT the funnel: two head coordinates (mu, log tau), rows theta_d ~ N(mu, tau^2) observed
# with unit noise at y_d = params[d]; exercises head = 2, the row index and params inside the row functions
HIER_LANES_SRC = """
template <class L>
__device__ double bk_lanes_density(L& c, const double* y) {
  const double mu = c.head(0), lt = c.head(1);
  const double it2 = exp(-2.0 * lt);  // 1 / tau^2
  const double n = (double)(c.dims() - 2);
  const double sq = c.sum([mu](double x, i64) { const double r = x - mu; return r * r; });
  const double sr = c.sum([mu](double x, i64) { return x - mu; });
  const double sy = c.sum([y](double x, i64 d) { const double r = y[d] - x; return r * r; });
  c.grad_head(0, it2 * sr - mu / 25.0);
  c.grad_head(1, (it2 * sq - n) - lt);
  c.grad([mu, it2, y](double x, i64 d) { return (y[d] - x) - it2 * (x - mu); });
  return (((-0.5 * it2) * sq - n * lt) - 0.5 * sy) - (mu * mu / 50.0 + 0.5 * (lt * lt));
}
"""


def funnel_lanes(D):
    return bk.CTarget.from_source(FUNNEL_LANES_SRC, D, form="lanes", head=1)


def funnel_plugin(D):
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return bk.CTarget(os.path.join(root, "examples", "plugin_target", "libfunnel_target.so"), "funnel_target", D,
                      counted_symbol="funnel_target_n")


def _same_state(a, b):
    return (torch.equal(a._theta_dc, b._theta_dc) and torch.equal(a._rho_dc, b._rho_dc)
            and torch.equal(a._rng_state, b._rng_state) and torch.equal(a._lp, b._lp))


def test_lanes_form_runs_the_one_launch_proposals_of_the_builtin_funnel(ops):
    """the funnel written as from_source(form="lanes") source goes through the SAME one-launch
    delayed-rejection proposal kernel template as bk.Funnel (csrc/bk_lanes.hpp) -- theta, rho, log density and RNG
    state bit-identical, no host synchronisation, one hipGraph per draw; with and without a metric, K = 1..4, shapes
    with 1..8 slots per class; through the reference goldens as well."""
    from tests.sampler_parity import check_many_chain

    cases = [(101, 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, None, 3000),
             (17, 4, [0.3, 0.15, 0.07, 0.03], [3, 6, 12, 24], 0.3, None, 700),
             (33, 2, [0.25, 0.1], [4, 8], 0.5, "metric", 1500),
             (129, 3, [0.2, 0.06, 0.02], [5, 15, 45], 0.2, None, 5000),
             (2, 2, [0.4, 0.2], [3, 6], 0.4, None, 300),
             (64, 1, [0.1], [7], 1.0, "metric", 20000)]
    for D, K, sizes, counts, damping, metric, C in cases:
        m = None if metric is None else np.linspace(0.5, 2.0, D)
        a = bk.DrGhmcDiag(bk.Funnel(D), K, sizes, counts, damping, metric_diag=m, chains=C, seed=11)
        b = bk.DrGhmcDiag(funnel_lanes(D), K, sizes, counts, damping, metric_diag=m, chains=C, seed=11)
        assert b._one_launch and b._dev_counts and b._use_graph and b.host_syncs_per_draw == 0, D
        for n in range(6):
            ta, la = a.sample()
            tb, lb = b.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), (D, n)
        assert _same_state(a, b), D
    for name in ("drghmc_funnel101_cfg4", "drghmc_funnel129_k3", "drghmc_funnel11_k3", "drghmc_funnel17_k4",
                 "drghmc_funnel33_k2_metric_noretry"):
        s = check_many_chain(name, ops, model_factory=lambda spec, ops_: funnel_lanes(spec["D"]))
        assert s._one_launch, name


def test_lanes_form_gradient_op_is_the_plugin_on_the_counted_path(ops):
    """the same source as a gradient OP (a chain spread over 4 / 8 / 16 lanes, DPP sums in the library's
    fixed order) on the counted step-by-step path: bit-identical to the hand-written plugin and to the one-launch path;
    past 128 spread rows (no one-launch kernel) the op walks the rows in memory, the plugin's class order."""
    from tests.sampler_parity import check_many_chain

    args = (3, [0.2, 0.05, 0.0125], [5, 10, 20], 0.1)
    for D, C, metric in ((101, 2500, None), (40, 13000, "m"), (130, 800, None), (300, 500, "m")):
        one = D - 1 <= 128
        kw = dict(chains=C, seed=21, metric_diag=None if metric is None else np.linspace(0.6, 1.7, D))
        a = bk.DrGhmcDiag(funnel_lanes(D), *args, path="opaque", **kw)  # gradient op per step
        b = bk.DrGhmcDiag(funnel_lanes(D), *args, path="step", **kw)  # {gradient, kick, drift} ONE launch per step
        h = bk.DrGhmcDiag(funnel_lanes(D), *args, path="step", device_counts=False, **kw)  # the same, host-sized
        p = bk.DrGhmcDiag(funnel_plugin(D), *args, **kw)
        f = bk.DrGhmcDiag(funnel_lanes(D), *args, **kw) if one else None
        assert a._dev_counts and a._use_graph and not a._one_launch and (f is None or f._one_launch)
        assert b._step_hook and h._step_hook and not a._step_hook and not p._step_hook and b._dev_counts and not h._dev_counts
        if not one:
            assert not bk.DrGhmcDiag(funnel_lanes(D), *args, chains=64, seed=1)._one_launch
        for n in range(5):
            ta, la = a.sample()
            tp, lpp = p.sample()
            assert torch.equal(ta, tp) and torch.equal(la, lpp), (D, n)
            for o in (b, h) + ((f,) if f is not None else ()):
                to, _ = o.sample()
                assert torch.equal(ta, to), (D, n)
        assert _same_state(a, p) and _same_state(a, b) and torch.equal(a._rng_state, h._rng_state), D
    # the op on its own: every geometry (lane count on the host / on the device, small / mid / large sets), logp-only calls
    for D in (101, 7, 300):
        for n in (1, 63, 4608, 12288 + 5):
            th = torch.randn((D, n), dtype=torch.float64, device=ops.device)
            out = []
            for model in (funnel_plugin(D), funnel_lanes(D)):
                g = torch.zeros_like(th)
                lp, lp2 = (torch.zeros(n, dtype=torch.float64, device=ops.device) for _ in range(2))
                model.bk_eval(th, g, lp)
                model.bk_eval(th, None, lp2)
                g_n = torch.zeros_like(th)
                nd = torch.tensor([max(1, n - 3)], dtype=torch.int32, device=ops.device)
                model.bk_eval(th, g_n, None, n_dev=nd)
                out.append((g, lp, lp2, g_n))
            for x, y in zip(*out):
                assert torch.equal(x, y), (D, n)
            assert torch.equal(out[1][3][:, n - 3:], torch.zeros_like(out[1][3][:, n - 3:])) or n <= 3
    # a reference golden through the counted path of the compiled source
    check_many_chain("drghmc_funnel130_k2", ops, model_factory=lambda spec, ops_: funnel_lanes(spec["D"]))


def test_lanes_form_hierarchical_model_with_two_head_coordinates(ops):
    """A density that is not the funnel (head = 2, three sums, params indexed by row): the one-launch path, the counted
    step-by-step path and the host-sized path agree bit for bit, and the compiled gradient is the autograd gradient of the
    same density written in PyTorch (rel 1e-12)."""
    D, C = 2 + 50, 1800
    y = torch.linspace(-2.0, 3.0, D, dtype=torch.float64, device=ops.device)
    mk = lambda: bk.CTarget.from_source(HIER_LANES_SRC, D, params=y, form="lanes", head=2)  # noqa: E731

    def torch_lp(Th):
        mu, lt, x = Th[:, 0], Th[:, 1], Th[:, 2:]
        it2 = torch.exp(-2.0 * lt)
        sq = ((x - mu[:, None]) ** 2).sum(dim=1)
        sy = ((y[2:] - x) ** 2).sum(dim=1)
        return -0.5 * it2 * sq - (D - 2) * lt - 0.5 * sy - (mu * mu / 50.0 + 0.5 * lt * lt)

    Th = torch.randn((C, D), dtype=torch.float64, device=ops.device) * 0.7
    lp, g = mk().log_density_gradient(Th)
    x = Th.clone().requires_grad_(True)
    lp_t = torch_lp(x)
    (g_t,) = torch.autograd.grad(lp_t.sum(), x)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_t.detach().cpu().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g.cpu().numpy(), g_t.cpu().numpy(), rtol=1e-11, atol=1e-11)
    args = (3, [0.15, 0.05, 0.02], [4, 8, 16], 0.2)
    m = np.linspace(0.7, 1.4, D)
    f = bk.DrGhmcDiag(mk(), *args, metric_diag=m, chains=C, seed=31)
    c = bk.DrGhmcDiag(mk(), *args, metric_diag=m, chains=C, seed=31, path="step")
    h = bk.DrGhmcDiag(mk(), *args, metric_diag=m, chains=C, seed=31, path="opaque", device_counts=False)
    assert f._one_launch and c._dev_counts and not c._one_launch and not h._dev_counts and c._step_hook and not h._step_hook
    for n in range(6):
        tf, lf = f.sample()
        tc, lc = c.sample()
        th_, lh = h.sample()
        assert torch.equal(tf, tc) and torch.equal(tf, th_), n
        np.testing.assert_allclose(lf.cpu().numpy(), lc.cpu().numpy(), rtol=1e-12, atol=1e-12)
    assert torch.equal(f._rng_state, c._rng_state) and torch.equal(f._rng_state, h._rng_state)
    assert torch.isfinite(f._theta_dc).all()


def test_elementwise_form_runs_the_whole_draw_hmc_kernel(ops):
    """a from_source(form="elementwise") density gets the register-resident whole-trajectory /
    whole-draw HMC kernels the built-in Gaussians have (csrc/bk_elementwise.hpp): one pass over the state per draw,
    bit-identical to its own step-by-step path and to the built-in target; reference goldens through it."""
    from tests.sampler_parity import check_many_chain

    for D, C, eps, L, metric in ((48, 1500, 0.02, 9, None), (1024, 256, 0.006, 16, "m"), (33, 4097, 0.05, 5, "m"), (5, 64, 0.1, 0, None)):
        lam = np.logspace(0, 2, D)
        lam_d = torch.from_numpy(lam).to(ops.device)
        m = None if metric is None else np.linspace(0.5, 2.0, D)
        src = lambda: bk.CTarget.from_source(DIAG_SRC, D, params=lam_d)  # noqa: E731
        kw = dict(metric_diag=m, chains=C, seed=3)
        f = bk.HMCDiag(src(), eps, L, **kw)
        s = bk.HMCDiag(src(), eps, L, path="step", **kw)
        b = bk.HMCDiag(bk.DiagGaussian(lam), eps, L, **kw)
        g = bk.HMCDiag(src(), eps, L, graph=True, **kw)
        assert f._fused_draw and not s._fused and b._fused_draw and g._fused_draw
        for n in range(6):
            tf, lf = f.sample()
            for o in (s, b, g):
                to, lo = o.sample()
                assert torch.equal(tf, to) and torch.equal(lf, lo), (D, n)
        np.testing.assert_array_equal(f.rng_state(), s.rng_state())
        assert f.accept_rate() == s.accept_rate()
    # PCG64 streams take the state-layout momentum (no chain-major generator): the same kernels' other input form
    f = bk.HMCDiag(bk.CTarget.from_source(DIAG_SRC, 48, params=torch.from_numpy(np.logspace(0, 2, 48)).to(ops.device)), 0.02, 9,
                   chains=40, seed=3)
    assert f._fused_draw

    def factory(spec, ops_):
        if spec["kind"] in ("iso_gaussian", "std_normal"):
            D_ = spec.get("D", 1)
            return bk.CTarget.from_source(DIAG_SRC, D_, params=torch.ones(D_, dtype=torch.float64, device=ops.device))
        lam_ = torch.from_numpy(np.logspace(spec["log10_lo"], spec["log10_hi"], spec["D"])).to(ops.device)
        return bk.CTarget.from_source(DIAG_SRC, spec["D"], params=lam_)

    for name in ("hmc_diag1024_cfg3", "hmc_iso128_cfg2", "hmc_diag40_metric_steps1", "hmc_steps0"):
        check_many_chain(name, ops, model_factory=factory)


def test_hmc_on_lane_spread_densities_one_launch_per_trajectory(ops):
    """HMC (hmc.py:40-63) on a lane-spread density -- bk.Funnel, the funnel and a two-head hierarchical model from source: the
    whole trajectory is ONE launch of the library's trajectory kernel with hmc.py's first kick (bk_hmc_proposal), and the
    step-by-step path issues ONE launch per leapfrog step (bk_leapfrog_step).  Both give the draws of the path with the
    gradient as a separate op per step, bit for bit (the returned joint log density to rounding: the fused kernel sums the
    kinetic energy in its lanes' order), and those are the oracle's HMC on the NumPy funnel."""
    from oracle import models as om
    from oracle import samplers as osamp
    from tests.sampler_parity import funnel_tol

    yv = torch.linspace(-2.0, 3.0, 52, dtype=torch.float64, device=ops.device)
    cases = [("funnel", lambda D: bk.Funnel(D), 101, 0.05, 12, None, 3000),
             ("funnel", lambda D: bk.Funnel(D), 17, 0.1, 1, "m", 700),
             ("funnel from source", funnel_lanes, 101, 0.05, 12, None, 3000),
             ("funnel from source", funnel_lanes, 33, 0.08, 7, "m", 20000),
             ("funnel, rows walked in memory", funnel_lanes, 300, 0.03, 5, None, 600),
             ("hierarchical", lambda D: bk.CTarget.from_source(HIER_LANES_SRC, D, params=yv, form="lanes", head=2), 52, 0.04, 9, "m", 1800)]
    for name, mk, D, eps, L, metric, C in cases:
        m = None if metric is None else np.linspace(0.7, 1.4, D)
        kw = dict(metric_diag=m, chains=C, seed=41)
        f = bk.HMCDiag(mk(D), eps, L, **kw)                                         # one launch per trajectory
        h = bk.HMCDiag(mk(D), eps, L, path="step", **kw)                     # one launch per leapfrog step
        s = bk.HMCDiag(mk(D), eps, L, path="opaque", **kw)   # gradient a separate op per step
        g = bk.HMCDiag(mk(D), eps, L, graph=True, **kw)
        one = D <= 129 or name == "hierarchical"
        assert f._lanes_traj == one and h._step_hook and not h._lanes_traj and not s._step_hook and not s._lanes_traj, name
        calls0 = s._grad_calls
        for n in range(8):
            tf, lf = f.sample()
            th_, lh = h.sample()
            ts, ls = s.sample()
            tg, lg = g.sample()
            assert torch.equal(th_, ts) and torch.equal(lh, ls), (name, D, n)
            assert torch.equal(tf, ts) and torch.equal(tg, ts), (name, D, n, one)
            torch.testing.assert_close(lf, ls, rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(f.rng_state(), s.rng_state())
        assert h._grad_calls == s._grad_calls, name  # (one model evaluation per leapfrog step on both paths)
        assert 0.2 < f.accept_rate() <= 1.0, (name, f.accept_rate())
    # the oracle's HMC on the NumPy funnel, chain by chain
    D, C, eps, L, seed = 21, 64, 0.1, 6, 77
    f = bk.HMCDiag(bk.Funnel(D), eps, L, chains=C, seed=seed)
    assert f._lanes_traj
    draws = [f.sample() for _ in range(10)]
    for c in range(0, C, 9):
        o = osamp.HMCDiag(om.Funnel(D), eps, L, seed=np.random.Philox(key=[seed, c]))
        for n in range(10):
            oth, olp = o.sample()
            np.testing.assert_allclose(draws[n][0][c].cpu().numpy(), oth, **funnel_tol(n))
            np.testing.assert_allclose(float(draws[n][1][c]), olp, **funnel_tol(n))


def test_hierarchical_example_converges():
    """examples/hierarchical_model.py end to end: a user's hierarchical model from source under DRGHMC (one launch per proposal)
    and HMC (one launch per trajectory), a PyTorch density traced and compiled: R-hat ~ 1, posterior where the data put it."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "hierarchical_model.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    txt = out.stdout
    assert "one launch per proposal: True | host syncs per draw: 0" in txt and "whole trajectory in one launch: True" in txt
    assert "compiled = True" in txt and "whole-draw kernel: True" in txt
    rh = [float(x) for x in re.search(r"R-hat: mu ([\d.]+)  log tau ([\d.]+)  max over rows ([\d.]+)", txt).groups()]
    assert max(rh) < 1.05, rh
    mu, ybar, tau, sd = [float(x) for x in re.search(r"mean of mu ([\d.]+) \(data mean ([\d.]+)\), of tau ([\d.]+) \(data sd ([\d.]+)\)", txt).groups()]
    assert abs(mu - ybar) < 0.05 and abs(tau - (sd * sd - 1.0) ** 0.5) < 0.1, (mu, ybar, tau, sd)


def test_state_space_example_converges():
    """examples/state_space_model.py: an AR(1) state-space model written with shifted slices in PyTorch, traced into the per-chain
    form, one launch per trajectory under DRGHMC: R-hat < 1.05, the posterior covers the truth, the path beats the observations."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "state_space_model.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    txt = out.stdout
    assert "compiled form = chain" in txt and "one launch per trajectory: True | host syncs per draw: 0" in txt
    assert "chains with a non-finite state: 0" in txt
    rh = [float(x) for x in re.search(r"R-hat: a ([\d.]+)  log s ([\d.]+)  max over states ([\d.]+)", txt).groups()]
    assert max(rh) < 1.05, rh
    phi, sd, s_ = [float(x) for x in re.search(r"mean of phi ([\d.]+) \(sd ([\d.]+); truth 0.8\), of s ([\d.]+)", txt).groups()]
    assert abs(phi - 0.8) < 3 * sd and abs(s_ - 0.5) < 0.15, (phi, sd, s_)
    assert float(re.search(r"true states ([\d.]+)", txt).group(1)) < 0.5


def test_torch_model_traces_hierarchical_densities_into_the_lanes_form(ops):
    """TorchModel(fn, D, compile=True) on head-plus-sums densities written in PyTorch (trace_lanes.py): Neal's funnel and a
    two-head hierarchical model become lane-spread compiled targets -- gradient equal to autograd, every DRGHMC proposal and
    every HMC trajectory ONE launch, the three DRGHMC paths bit-identical among themselves, and the traced funnel tracks
    bk.Funnel (the same density in another rounding) over the first draws."""
    D = 101

    def funnel(Th):
        v, x = Th[:, 0], Th[:, 1:]
        return -(v * v) / 18.0 - 0.5 * (D - 1) * v - 0.5 * torch.exp(-v) * (x * x).sum(dim=1)

    Dh = 52
    yv = torch.linspace(-2.0, 3.0, Dh, dtype=torch.float64, device=ops.device)[2:]

    def hier(Th):
        mu, lt, x = Th[:, 0], Th[:, 1], Th[:, 2:]
        it2 = torch.exp(-2.0 * lt)
        sq = ((x - mu[:, None]) ** 2).sum(dim=1)
        sy = ((yv - x) ** 2).sum(dim=1)
        return -0.5 * it2 * sq - (Dh - 2) * lt - 0.5 * sy - (mu * mu / 50.0 + 0.5 * lt * lt)

    for fn, dims, head in ((funnel, D, 1), (hier, Dh, 2)):
        m = bk.TorchModel(fn, dims, compile=True)
        assert m.compiled is not None and m.compiled_form == "lanes" and m.compiled._head == head, m.compile_note
        Th = 0.7 * torch.randn((777, dims), dtype=torch.float64, device=ops.device)
        x = Th.clone().requires_grad_(True)
        lp_t = fn(x)
        (g_t,) = torch.autograd.grad(lp_t.sum(), x)
        lp, g = m.log_density_gradient(Th)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_t.detach().cpu().numpy(), rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(g.cpu().numpy(), g_t.cpu().numpy(), rtol=1e-10, atol=1e-11 * float(g_t.abs().max()))
        args = (3, [0.15, 0.05, 0.02], [4, 8, 16], 0.2)
        f = bk.DrGhmcDiag(m, *args, chains=1500, seed=51)
        c = bk.DrGhmcDiag(bk.TorchModel(fn, dims, compile=True), *args, chains=1500, seed=51, path="step")
        hs = bk.DrGhmcDiag(bk.TorchModel(fn, dims, compile=True), *args, chains=1500, seed=51, path="opaque",
                           device_counts=False)
        assert f._one_launch and f.host_syncs_per_draw == 0 and c._step_hook and not c._one_launch and not hs._dev_counts
        for n in range(6):
            tf, _ = f.sample()
            tc, _ = c.sample()
            th_, _ = hs.sample()
            assert torch.equal(tf, tc) and torch.equal(tf, th_), (dims, n)
        hm = bk.HMCDiag(bk.TorchModel(fn, dims, compile=True), 0.05, 8, chains=1500, seed=52)
        assert hm._lanes_traj
        for _ in range(4):
            th, lp = hm.sample()
        assert torch.isfinite(th).all() and 0.3 < hm.accept_rate() <= 1.0
    # the traced funnel against the built-in one: the same density, another rounding of the gradient
    a = bk.DrGhmcDiag(bk.Funnel(D), 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=512, seed=20242)
    b = bk.DrGhmcDiag(bk.TorchModel(funnel, D, compile=True), 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=512, seed=20242)
    for n in range(4):
        ta, la = a.sample()
        tb, lb = b.sample()
        np.testing.assert_allclose(tb.cpu().numpy(), ta.cpu().numpy(), rtol=1e-8, atol=1e-9)
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())

@pytest.mark.gpu
def test_torch_model_traces_coupled_densities_into_the_per_chain_form(ops):
    """TorchModel(fn, D, compile=True) on densities whose coordinates are coupled through shifted slices (trace_chain.py): an AR(1)
    state-space model with learned correlation and scale, a second-order random-walk prior, a stochastic-volatility model --
    compiled into the per-chain form: gradient equal to autograd, one launch per trajectory == one launch per step == gradient a
    separate op under DRGHMC (counted and host-sized) and HMC, bit for bit."""
    dev = ops.device
    D = 101
    y = torch.sin(torch.linspace(0.0, 9.0, D - 2, dtype=torch.float64, device=dev))
    y1 = torch.cat([y, y[:1]])

    def ar1(Th):
        phi, ls, x = torch.tanh(Th[:, 0]), Th[:, 1], Th[:, 2:]
        inn = x[:, 1:] - phi[:, None] * x[:, :-1]
        return -0.5 * (inn * inn).sum(-1) * torch.exp(-2 * ls) - (D - 3) * ls - 0.5 * x[:, 0] ** 2 * (1 - phi * phi) * torch.exp(-2 * ls) \
            - 2.0 * ((y - x) ** 2).sum(-1) - 0.5 * Th[:, 0] ** 2 - 0.5 * (ls + 1.0) ** 2 / 0.09

    def rw2(Th):
        d2 = torch.diff(torch.diff(Th, dim=1), dim=1)
        return -0.5 * (d2 ** 2).sum(1) * 4.0 - 0.05 * (Th ** 2).sum(1)

    def sv(Th):
        mu, h = Th[:, 0], Th[:, 1:]
        return -0.5 * ((h[:, 1:] - mu[:, None] - 0.9 * (h[:, :-1] - mu[:, None])) ** 2).sum(-1) / 0.04 \
            - 0.5 * (h + y1 * y1 * torch.exp(-h)).sum(-1) - 0.5 * mu * mu

    for fn, dims in ((ar1, D), (rw2, 64), (sv, D), (rw2, 150)):
        m = bk.TorchModel(fn, dims, compile=True)
        assert m.compiled is not None and m.compiled_form == "chain", (fn.__name__, m.compile_note)
        Th = 0.3 * torch.randn((777, dims), dtype=torch.float64, device=dev)
        x = Th.clone().requires_grad_(True)
        lp_t = fn(x)
        (g_t,) = torch.autograd.grad(lp_t.sum(), x)
        lp, g = m.log_density_gradient(Th)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_t.detach().cpu().numpy(), rtol=1e-11, atol=1e-10, err_msg=fn.__name__)
        np.testing.assert_allclose(g.cpu().numpy(), g_t.cpu().numpy(), rtol=1e-10, atol=1e-11 * float(g_t.abs().max()),
                                   err_msg=fn.__name__)
        args = (3, [0.03, 0.012, 0.005], [4, 8, 16], 0.2)
        th0 = 0.3 * torch.randn((1500, dims), dtype=torch.float64, device=dev)
        mk = lambda **kw: bk.DrGhmcDiag(bk.TorchModel(fn, dims, compile=True), *args, chains=1500, seed=71, init=th0, **kw)  # noqa: E731
        f, c, op, hs = mk(), mk(path="step"), mk(path="opaque"), mk(device_counts=False)
        assert f._traj_hook == (dims <= 128) and c._step_hook == (dims <= 128) and not c._traj_hook and not op._step_hook
        assert f._dev_counts and not hs._dev_counts and f.host_syncs_per_draw == 0
        for n in range(5):
            tf, lf = f.sample()
            assert torch.isfinite(tf).all(), (fn.__name__, n)
            for other in (c, op, hs):
                to, lo = other.sample()
                assert torch.equal(tf, to) and torch.equal(lf, lo), (fn.__name__, dims, n)
        ha = bk.HMCDiag(bk.TorchModel(fn, dims, compile=True), 0.02, 8, chains=1500, seed=72, init=th0)
        hb = bk.HMCDiag(bk.TorchModel(fn, dims, compile=True), 0.02, 8, chains=1500, seed=72, init=th0, path="opaque")
        assert ha._traj_hook == (dims <= 128)
        for _ in range(4):
            ta, la = ha.sample()
            tb, lb = hb.sample()
            assert torch.equal(ta, tb) and torch.equal(la, lb), fn.__name__
        assert 0.3 < ha.accept_rate() <= 1.0, (fn.__name__, ha.accept_rate())


@pytest.mark.gpu
def test_torch_model_traces_distributions_row_groups_and_piecewise_densities(ops):
    """The wider traced shapes on the GPU: a funnel written with torch.distributions log_prob calls, row GROUPS (several slices
    with their own hyper-parameters), a density with NO head coordinate but a nonlinear function of its sums, piecewise rows
    (where / relu / clamp / maximum) and a separable sum of distribution bodies: gradient equal to autograd, DRGHMC one launch
    per proposal bit-identical to the step-by-step paths, HMC sane."""
    Normal, Laplace = torch.distributions.Normal, torch.distributions.Laplace
    dev = ops.device
    w = torch.linspace(0.5, 1.5, 40, dtype=torch.float64, device=dev)

    def funnel_dist(Th):
        v, x = Th[:, 0], Th[:, 1:]
        return Normal(0.0, 3.0).log_prob(v) + Normal(0.0, torch.exp(0.5 * v)[:, None]).log_prob(x).sum(-1)

    def groups(Th):
        mu, lt = Th[:, 0], Th[:, 1]
        ga, gb, gc = Th[:, 2:30], Th[:, 30:50], Th[:, 50:]
        la = Normal(mu[:, None], torch.exp(lt)[:, None]).log_prob(ga).sum(-1)
        lb = Laplace(0.0, 2.0).log_prob(gb - mu.unsqueeze(1)).sum(-1)
        lc = -0.5 * (w * gc * gc).sum(-1) + 0.3 * torch.tanh(gc).mean(-1) * lt
        return la + lb + lc - 0.5 * (mu * mu + lt * lt / 0.04)  # (a tight prior on the log scale: no funnel neck to fall into)

    def no_heads(Th):
        return -0.5 * (Th ** 2).sum(-1) - torch.log(torch.exp(Th).sum(-1))

    def piecewise(Th):
        s, x = Th[:, 0], Th[:, 1:]
        z = x * torch.exp(-s)[:, None]
        return (torch.where(z > 0.0, -z, 2.0 * z) - torch.relu(z - 0.5) ** 2 - torch.clamp(x, -0.4, 0.6) ** 2).sum(1) \
            - 0.5 * s * s - 32.0 * torch.maximum(s, 0.1 * s)

    def bodies(Th):
        return (torch.distributions.StudentT(4.0).log_prob(Th) + torch.distributions.Gumbel(0.0, 1.0).log_prob(Th)
                - torch.log(torch.cosh(0.5 * Th)) + 0.1 * torch.erf(Th)).sum(-1)

    for fn, dims, head, form in ((funnel_dist, 101, 1, "lanes"), (groups, 90, 2, "lanes"), (no_heads, 64, 0, "lanes"),
                                 (piecewise, 33, 1, "lanes"), (bodies, 48, None, "elementwise")):
        m = bk.TorchModel(fn, dims, compile=True)
        assert m.compiled is not None and m.compiled_form == form, (fn.__name__, m.compile_note)
        if head is not None:
            assert m.compiled._head == head
        Th = 0.7 * torch.randn((777, dims), dtype=torch.float64, device=dev)
        x = Th.clone().requires_grad_(True)
        lp_t = fn(x)
        (g_t,) = torch.autograd.grad(lp_t.sum(), x)
        lp, g = m.log_density_gradient(Th)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_t.detach().cpu().numpy(), rtol=1e-11, atol=1e-10, err_msg=fn.__name__)
        np.testing.assert_allclose(g.cpu().numpy(), g_t.cpu().numpy(), rtol=1e-10, atol=1e-11 * float(g_t.abs().max()),
                                   err_msg=fn.__name__)
        args = (3, [0.06, 0.025, 0.01], [4, 8, 16], 0.2)
        th0 = 0.5 * torch.randn((1500, dims), dtype=torch.float64, device=dev)
        f = bk.DrGhmcDiag(m, *args, chains=1500, seed=61, init=th0)
        hs = bk.DrGhmcDiag(bk.TorchModel(fn, dims, compile=True), *args, chains=1500, seed=61, path="opaque",
                           device_counts=False, init=th0)
        assert f._one_launch and f.host_syncs_per_draw == 0 and not hs._one_launch
        for n in range(5):
            tf, _ = f.sample()
            th_, _ = hs.sample()
            assert torch.isfinite(tf).all(), (fn.__name__, n)
            assert torch.equal(tf, th_), (fn.__name__, n)
        hm = bk.HMCDiag(bk.TorchModel(fn, dims, compile=True), 0.03, 8, chains=1500, seed=62, init=th0)
        for _ in range(4):
            th, lp = hm.sample()
        assert torch.isfinite(th).all() and 0.3 < hm.accept_rate() <= 1.0, (fn.__name__, hm.accept_rate())




CHAIN_GOOD = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* lam) {
  double s = 0.0;
  for (i64 d = 0; d < D; ++d) { const double t = lam[d] * th[d]; s = s + th[d] * t; if (g.wanted()) g.set(d, -t); }
  return -0.5 * s;
}"""
CHAIN_SETS_TWICE = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* lam) {
  double s = 0.0;
  if (g.wanted()) for (i64 d = 0; d < D; ++d) g.set(d, -(0.5 * lam[d]) * th[d]);   // "a first part, overwritten below": breaks the contract
  for (i64 d = 0; d < D; ++d) { const double t = lam[d] * th[d]; s = s + th[d] * t; if (g.wanted()) g.set(d, -t); }
  return -0.5 * s;
}"""
CHAIN_SKIPS_ONE = """
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double* lam) {
  double s = 0.0;
  for (i64 d = 0; d < D; ++d) { const double t = lam[d] * th[d]; s = s + th[d] * t; if (g.wanted() && (d != 3 || t != 0.0)) g.set(d, -t); }
  return -0.5 * s;   // (entry 3 is skipped when its gradient is zero: "nothing to add")
}"""


def test_chain_form_step_kernels_are_checked_against_the_gradient_op_when_the_object_is_built(ops):
    """in the one-launch step / trajectory kernels of form="chain" BkGrad::set is a kick, not a store.
    from_source runs both against {gradient op, kick + drift} on random points: a function that keeps the contract keeps
    its kernels; one that sets entries twice loses them (with a warning) and the samplers fall back to the separate
    gradient op -- same draws as the well-behaved source, bit for bit."""
    import warnings

    D = 24
    lam = torch.linspace(0.5, 3.0, D, dtype=torch.float64, device=ops.device)
    good = bk.CTarget.from_source(CHAIN_GOOD, D, params=lam, form="chain")
    assert good.chain_hooks_note.startswith("checked") and hasattr(good, "bk_leapfrog_step") and hasattr(good, "bk_leapfrog_trajectory")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        twice = bk.CTarget.from_source(CHAIN_SETS_TWICE, D, params=lam, form="chain")
    assert any("exactly once" in str(x.message) for x in w)
    assert twice.chain_hooks_note.startswith("dropped") and "bk_leapfrog_trajectory" in twice.chain_hooks_note
    assert not hasattr(twice, "bk_leapfrog_step") and not hasattr(twice, "bk_leapfrog_trajectory")
    # (the skipping variant keeps the contract at the random test points -- no gradient entry is exactly zero there -- and is
    # indistinguishable from a correct function: the check is a net for systematic violations, the contract is the documentation)
    skip = bk.CTarget.from_source(CHAIN_SKIPS_ONE, D, params=lam, form="chain")
    assert skip.chain_hooks_note.startswith("checked")
    # the dropped hooks change the route, not the draws
    mk = lambda m: bk.DrGhmcDiag(m, 2, [0.3, 0.1], [3, 6], 0.3, chains=500, seed=11)   # noqa: E731
    a, b = mk(good), mk(twice)
    for _ in range(6):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert torch.equal(ta, tb) and torch.equal(la, lb)
    h1 = bk.HMCDiag(good, 0.1, 7, chains=333, seed=5)
    h2 = bk.HMCDiag(twice, 0.1, 7, chains=333, seed=5)
    for _ in range(4):
        assert torch.equal(h1.sample()[0], h2.sample()[0])
