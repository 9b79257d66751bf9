"""User-side models for the provider x sampler parity tests: the three ways a user hands a log density
to the engine (bayes_kit/typing.py:25-27 in its batched form) other than the library's own targets --
PyTorch autograd (``bk.TorchModel``), hand-written batched PyTorch code returning the gradient in a layout
of its own, and the compiled plugin (``bk.CTarget``).  Each has a single-chain NumPy twin under ``oracle/`` or
``tests/host_models.py`` that the oracle samplers drive."""
import ctypes
import os

import numpy as np
import torch

import bayes_kit_amd as bk

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def torch_diag_gaussian(lam, device):
    """logp = -1/2 sum lam_i theta_i^2 through autograd.  The backward pass forms (-0.5*lam)*theta twice and
    adds the two: scalings by powers of two commute with rounding, so the gradient is -(lam*theta) bit for
    bit -- what oracle.models.DiagGaussian returns."""
    lam_t = torch.as_tensor(lam, dtype=torch.float64, device=device)
    return bk.TorchModel(lambda Th: -0.5 * ((Th * Th) * lam_t).sum(dim=1), lam_t.shape[0])


def torch_funnel(D):
    """Neal's funnel written in torch ops (oracle.models.Funnel's density): exp + a reduction inside the
    gradient, which autograd differentiates -- d/dv comes out as -(2v)/18 where the oracle writes -v/9."""
    hn = 0.5 * (D - 1)

    def fn(Th):
        v, x = Th[:, 0], Th[:, 1:]
        s = (x * x).sum(dim=1)
        return ((-(v * v) / 18.0) - hn * v) - (0.5 * torch.exp(-v)) * s

    return bk.TorchModel(fn, D)


def torch_logistic(X, y, prior_scale, device):
    """Bayesian logistic regression in torch ops (oracle.models.LogisticRegression's density)."""
    Xt = torch.as_tensor(X, dtype=torch.float64, device=device)
    yt = torch.as_tensor(y, dtype=torch.float64, device=device)
    inv_s2 = 1.0 / prior_scale ** 2

    def fn(Th):
        z = Th @ Xt.t()
        return (yt * z - torch.nn.functional.softplus(z)).sum(dim=1) - 0.5 * inv_s2 * (Th * Th).sum(dim=1)

    return bk.TorchModel(fn, Xt.shape[1])


class RowMajorDiag:
    """A hand-written batched model whose gradient comes back as a fresh ROW-MAJOR (C, D) tensor
    (dimension-contiguous: the layout PyTorch code produces by default), or, with ``layout="strided"``, as a
    view into a wider buffer (neither chain- nor dimension-contiguous rows of the engine's kind)."""

    batched = True

    def __init__(self, lam, device, layout="row"):
        self._lam = torch.as_tensor(lam, dtype=torch.float64, device=device)
        self._layout = layout

    def dims(self):
        return self._lam.shape[0]

    def log_density(self, Th):
        return -0.5 * ((Th * Th) * self._lam).sum(dim=1)

    def log_density_gradient(self, Th):
        g = -(self._lam * Th)
        if self._layout == "row":
            g = g.contiguous()
            assert g.stride(1) == 1 or g.shape[1] == 1
        else:
            wide = torch.zeros((g.shape[0], 2 * g.shape[1] + 3), dtype=torch.float64, device=g.device)
            wide[:, 1::2][:, :g.shape[1]] = g
            g = wide[:, 1::2][:, :g.shape[1]]
        return self.log_density(Th), g


class Ar1Params(ctypes.Structure):
    _fields_ = [("a", ctypes.c_double), ("s2", ctypes.c_double)]


def ar1_plugin(D, a, s2):
    """examples/plugin_target/libar1_target.so through bk.CTarget (plugin ABI bk_target_fn)."""
    return bk.CTarget(os.path.join(ROOT, "examples", "plugin_target", "libar1_target.so"), "ar1_target", D,
                      Ar1Params(a, s2))


def compare_with_oracle(dev_sampler, make_oracle, draws, chains, seed, exact, tol=None, logp_tol=None):
    """`draws` draws of a many-chain device sampler against one oracle sampler per watched chain
    (``make_oracle(np.random.Philox(key=[seed, c]))``): theta bit for bit (``exact``) or within ``tol(n)``,
    returned log density within ``logp_tol``, and the chain's stream exactly where NumPy's ends."""
    from tests.helpers import rng_state_words

    th0 = np.array(dev_sampler._theta.cpu().numpy())  # (a copy: on CPU ops .cpu() aliases the state)
    outs = [tuple(np.array(x.cpu().numpy()) for x in dev_sampler.sample()) for _ in range(draws)]
    state = dev_sampler.rng_state()
    worst = 0.0
    for c in chains:
        o = make_oracle(np.random.Philox(key=[seed, c]))
        assert np.array_equal(np.asarray(o._theta), th0[c]), ("theta0", c)
        for n, (th, lp) in enumerate(outs):
            oth, olp = o.sample()
            if exact:
                assert np.array_equal(th[c], oth), (type(dev_sampler).__name__, "chain", c, "draw", n,
                                                    float(np.abs(th[c] - oth).max()))
                np.testing.assert_allclose(lp[c], olp, **(logp_tol or dict(rtol=1e-11, atol=1e-12)))
            else:
                t = tol(n)
                worst = max(worst, float((np.abs(th[c] - oth) / (t["atol"] + t["rtol"] * np.abs(oth))).max()))
                np.testing.assert_allclose(th[c], oth, err_msg=f"chain {c} draw {n}", **t)
                np.testing.assert_allclose(lp[c], olp, err_msg=f"chain {c} draw {n}", **(logp_tol or t))
        np.testing.assert_array_equal(state[:, c], rng_state_words(o._rng), err_msg=f"stream of chain {c}")
    if not exact:
        print(f"{type(dev_sampler).__name__}: worst theta error / allowed = {worst:.3g} over {len(chains)} chains x {draws} draws")
    return outs


# ---------------------------------------------------------------------------------------------------------
# test bodies (run with the HIP ops on the GPU and, but for the compiled plugin, with tests/fake_ops.py here)
# ---------------------------------------------------------------------------------------------------------
DR3 = (3, [0.3, 0.1, 0.03], [3, 6, 12], 0.3)
DR2 = (2, [0.4, 0.15], [2, 5], 0.5)


def _watch(C):
    return sorted({0, 1, C // 3, C // 2, C - 2, C - 1})


def check_torch_diag_gaussian(ops, C=301, D=16, draws=8, seed=4101):
    """(a) autograd DiagGaussian under MALA, DRGHMC K = 3 and K = 2 without probabilistic retry: theta bit
    for bit the oracle's (mala.py:46-48, drghmc.py:280-288 consume the model's gradient as it comes)."""
    from oracle import models as om
    from oracle import samplers as osamp

    lam = np.logspace(0, 1, D)
    mk = lambda: torch_diag_gaussian(lam, ops.device)  # noqa: E731
    compare_with_oracle(bk.MALA(mk(), 0.02, chains=C, seed=seed, ops=ops),
                        lambda sd: osamp.MALA(om.DiagGaussian(lam), 0.02, seed=sd), draws, _watch(C), seed, True)
    s = bk.DrGhmcDiag(mk(), *DR3, chains=C, seed=seed + 1, ops=ops)
    assert not s._fused and not s._dev_counts
    compare_with_oracle(s, lambda sd: osamp.DrGhmcDiag(om.DiagGaussian(lam), *DR3, seed=sd), draws, _watch(C),
                        seed + 1, True)
    stages = {t for t, _ in s.last_stage_lanes}
    s2 = bk.DrGhmcDiag(mk(), *DR2, chains=C, seed=seed + 2, prob_retry=False, ops=ops)
    s2._metric = np.linspace(0.9, 1.1, D)
    compare_with_oracle(s2, lambda sd: osamp.DrGhmcDiag(om.DiagGaussian(lam), *DR2, seed=sd, prob_retry=False,
                                                        metric_diag=np.linspace(0.9, 1.1, D)),
                        draws, _watch(C), seed + 2, True)
    return stages


def check_torch_funnel(ops, C=200, D=11, draws=10, seed=4201, cfg4_steps=True):
    """(b) Neal's funnel written in torch ops, differentiated by autograd (a transcendental user model),
    under all three samplers against oracle.models.Funnel: the funnel tolerance schedule of
    tests/sampler_parity.py, every decision agreeing (stream states equal)."""
    from oracle import models as om
    from oracle import samplers as osamp
    from tests.sampler_parity import funnel_tol

    compare_with_oracle(bk.HMCDiag(torch_funnel(D), 0.05, 6, chains=C, seed=seed, ops=ops),
                        lambda sd: osamp.HMCDiag(om.Funnel(D), 0.05, 6, seed=sd), draws, _watch(C), seed, False,
                        tol=funnel_tol)
    compare_with_oracle(bk.MALA(torch_funnel(D), 0.01, chains=C, seed=seed + 1, ops=ops),
                        lambda sd: osamp.MALA(om.Funnel(D), 0.01, seed=sd), draws, _watch(C), seed + 1, False,
                        tol=funnel_tol)
    args = (3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1) if cfg4_steps else DR3
    s = bk.DrGhmcDiag(torch_funnel(D), *args, chains=C, seed=seed + 2, ops=ops)
    compare_with_oracle(s, lambda sd: osamp.DrGhmcDiag(om.Funnel(D), *args, seed=sd), draws, _watch(C), seed + 2,
                        False, tol=funnel_tol)
    s2 = bk.DrGhmcDiag(torch_funnel(D), *DR2, chains=C, seed=seed + 3, prob_retry=False, ops=ops)
    compare_with_oracle(s2, lambda sd: osamp.DrGhmcDiag(om.Funnel(D), *DR2, seed=sd, prob_retry=False), draws,
                        _watch(C), seed + 3, False, tol=funnel_tol)
    return {t for t, _ in s.last_stage_lanes}


def check_torch_logistic(ops, N=300, D=12, C=130, draws=6, seed=4301):
    """(b) a small Bayesian logistic regression in torch ops under the three samplers against
    oracle.models.LogisticRegression: rel 1e-9 (a matmul and a softplus inside the gradient)."""
    from oracle import models as om
    from oracle import samplers as osamp

    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D)) / np.sqrt(D)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(np.float64)
    omodel = om.LogisticRegression(X, y, prior_scale=2.0)
    mk = lambda: torch_logistic(X, y, 2.0, ops.device)  # noqa: E731
    tol = lambda n: dict(rtol=1e-9, atol=1e-11)  # noqa: E731
    compare_with_oracle(bk.HMCDiag(mk(), 0.05, 5, chains=C, seed=seed, ops=ops),
                        lambda sd: osamp.HMCDiag(omodel, 0.05, 5, seed=sd), draws, _watch(C), seed, False, tol=tol)
    compare_with_oracle(bk.MALA(mk(), 0.01, chains=C, seed=seed + 1, ops=ops),
                        lambda sd: osamp.MALA(omodel, 0.01, seed=sd), draws, _watch(C), seed + 1, False, tol=tol)
    compare_with_oracle(bk.DrGhmcDiag(mk(), *DR3, chains=C, seed=seed + 2, ops=ops),
                        lambda sd: osamp.DrGhmcDiag(omodel, *DR3, seed=sd), draws, _watch(C), seed + 2, False, tol=tol)


def check_gradient_layouts(ops, C=333, D=40, draws=8, seed=4401):
    """(d) user gradients in layouts of their own -- row-major (C, D) and a strided view -- under MALA and
    DRGHMC: the compacted lane sets of delayed rejection ([D, n] views with leading dimension C != n) go
    through the LDS-transposing kick+drift (k_kick_drift_tr) and bk_relayout.  Bit for bit the oracle."""
    from oracle import models as om
    from oracle import samplers as osamp

    lam = np.logspace(0, 1, D)
    for layout in ("row", "strided"):
        mk = lambda: RowMajorDiag(lam, ops.device, layout)  # noqa: E731
        compare_with_oracle(bk.HMCDiag(mk(), 0.05, 5, chains=C, seed=seed, ops=ops),
                            lambda sd: osamp.HMCDiag(om.DiagGaussian(lam), 0.05, 5, seed=sd), draws, _watch(C), seed, True)
        compare_with_oracle(bk.MALA(mk(), 0.02, chains=C, seed=seed + 1, ops=ops),
                            lambda sd: osamp.MALA(om.DiagGaussian(lam), 0.02, seed=sd), draws, _watch(C), seed + 1, True)
        s = bk.DrGhmcDiag(mk(), *DR3, chains=C, seed=seed + 2, ops=ops)
        s._metric = np.linspace(0.8, 1.2, D)
        compare_with_oracle(s, lambda sd: osamp.DrGhmcDiag(om.DiagGaussian(lam), *DR3, seed=sd,
                                                           metric_diag=np.linspace(0.8, 1.2, D)),
                            draws, _watch(C), seed + 2, True)
        lanes = dict(s.last_stage_lanes)
        assert lanes["P0"] == C


def check_plugin_target(ops, C=257, D=16, draws=8, seed=4501):
    """(c) the compiled AR(1) plugin (bk.CTarget) under MALA and DRGHMC against its NumPy statement
    (tests/host_models.Ar1) driven by the oracle samplers: bit for bit."""
    from oracle import samplers as osamp
    from tests.host_models import Ar1

    a, s2 = 0.6, 0.8
    host = lambda: Ar1(D, a, s2)  # noqa: E731
    compare_with_oracle(bk.MALA(ar1_plugin(D, a, s2), 0.02, chains=C, seed=seed, ops=ops),
                        lambda sd: osamp.MALA(host(), 0.02, seed=sd), draws, _watch(C), seed, True)
    s = bk.DrGhmcDiag(ar1_plugin(D, a, s2), *DR3, chains=C, seed=seed + 1, ops=ops)
    compare_with_oracle(s, lambda sd: osamp.DrGhmcDiag(host(), *DR3, seed=sd), draws, _watch(C), seed + 1, True)
    s2_ = bk.DrGhmcDiag(ar1_plugin(D, a, s2), *DR2, chains=C, seed=seed + 2, prob_retry=False, ops=ops)
    compare_with_oracle(s2_, lambda sd: osamp.DrGhmcDiag(host(), *DR2, seed=sd, prob_retry=False), draws, _watch(C),
                        seed + 2, True)
    return {t for t, _ in s.last_stage_lanes}
