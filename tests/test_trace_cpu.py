"""bayes_kit_amd.trace on the CPU: the bk_term source generated from a PyTorch log density (value + hand-differentiated
derivative) is compiled here AS HOST C++ (g++; the generated text uses nothing device-specific) and compared with the
function itself and with torch autograd.  The GPU tests run the same source through CTarget.from_source."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import bayes_kit_amd as bk
from bayes_kit_amd import trace

F = torch.nn.functional
_BKHIP_MATH_H = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "bkhip_math.h")   # bk_exp
HOST = """
#include <math.h>
#include <stdint.h>
#include "BKHIP_MATH_H_PATH"
typedef int64_t i64;
#define __device__
#define __forceinline__ inline
%s
extern "C" void eval_terms(const double* th, const long long* d, const double* P, long long n, double* term, double* grad) {
  for (long long i = 0; i < n; ++i) bk_term(th[i], d[i], P, term[i], grad[i]);
}
"""


def host_eval(src, params, Theta, tmp_path, tag):
    """log density (C,) and gradient (C, D) of the generated bk_term, evaluated on the host."""
    cpp, lib = tmp_path / f"{tag}.cpp", tmp_path / f"lib{tag}.so"
    cpp.write_text((HOST % src).replace("BKHIP_MATH_H_PATH", _BKHIP_MATH_H))
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", str(cpp), "-o", str(lib)])
    h = ctypes.CDLL(str(lib))
    C, D = Theta.shape
    th = np.ascontiguousarray(Theta.numpy().reshape(-1))
    d = np.ascontiguousarray(np.tile(np.arange(D, dtype=np.int64), C))
    P = np.zeros(1) if params is None else np.ascontiguousarray(params.numpy())
    term, grad = np.empty(C * D), np.empty(C * D)
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    h.eval_terms(as_p(th), as_p(d), as_p(P), ctypes.c_longlong(C * D), as_p(term), as_p(grad))
    return term.reshape(C, D).sum(axis=1), grad.reshape(C, D)


D = 24
g = torch.Generator().manual_seed(7)
lam = torch.logspace(0, 2, D, dtype=torch.float64)
a = torch.randn(D, generator=g, dtype=torch.float64)
b = torch.rand(D, generator=g, dtype=torch.float64) + 0.5
nu = 4.0

DENSITIES = {
    "config3_gaussian": (lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), "cd"),
    "logistic_prior": (lambda Th: (-F.softplus(-a * Th) - 0.5 * Th ** 2 / 2.25).sum(dim=1), "cd"),
    "student_t": (lambda Th: (-(nu + 1.0) / 2.0 * torch.log1p(Th.pow(2) / (nu * b))).sum(-1), "cd"),
    "mixed_ops": (lambda Th: (torch.sigmoid(b * Th) * torch.exp(-(Th - a).abs()) + torch.tanh(Th) / (1.0 + Th.square())
                              - torch.expm1(-Th * Th)).sum(1) * 0.25 - 3.0, "cd"),
    "engine_layout_two_sums": (lambda Th: 2.0 * (F.logsigmoid(Th * a[:, None])).sum(dim=0)
                               - (torch.sqrt(1.0 + Th * Th) * lam[:, None]).sum(0) / 4.0 + 1.5, "dc"),
    "powers": (lambda Th: (-(1.0 + Th * Th) ** 1.5 + 0.1 * Th ** 3 - 2.0 ** (0.3 * Th) + torch.sin(Th) * torch.cos(b * Th)).sum(1),
               "cd"),
    "distributions_bodies": (lambda Th: (torch.distributions.Normal(0.5, 2.0).log_prob(Th)
                                         + torch.distributions.StudentT(4.0).log_prob(Th)
                                         + torch.distributions.Cauchy(0.0, 1.5).log_prob(Th)
                                         + torch.distributions.Laplace(0.25, 2.0).log_prob(Th)
                                         + torch.distributions.Gumbel(0.0, 1.0).log_prob(Th)
                                         + torch.distributions.LogNormal(0.0, 1.0).log_prob(torch.exp(Th))).sum(-1), "cd"),
    "piecewise": (lambda Th: (torch.where(Th > a, -0.5 * Th * Th, -torch.abs(Th) * b) - F.relu(Th - 1.0) ** 2
                              - torch.clamp(Th, -0.7, 0.9) ** 2 - Th.clamp(min=0.2) * 0.5 + torch.minimum(Th, a * Th) * 0.1
                              - torch.maximum(Th * Th, b)).sum(1), "cd"),
    "hyperbolic_erf": (lambda Th: (-torch.log(torch.cosh(Th)) + 0.1 * torch.sinh(0.5 * Th) + torch.atan(Th * b) * a
                                   + torch.erf(Th) * 0.3 + torch.rsqrt(1.0 + Th * Th) + torch.reciprocal(2.0 + Th.square())).mean(1)
                       * Th.shape[1], "cd"),
    "log_and_division": (lambda Th: (torch.log(1.0 + torch.exp(Th)) / (2.0 + torch.cos(Th)) - (a - Th) / b).sum(dim=1).neg(), "cd"),
}


@pytest.mark.parametrize("name", sorted(DENSITIES))
def test_generated_term_and_derivative_match_autograd(name, tmp_path):
    fn, layout = DENSITIES[name]
    src, params, info = trace.term_source(fn, D, layout)
    Theta = torch.randn((40, D), generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    x = (Theta.t() if layout == "dc" else Theta).clone().requires_grad_(True)
    lp = fn(x)
    (gr,) = torch.autograd.grad(lp.sum(), x)
    gr = gr.t() if layout == "dc" else gr
    lp_c, g_c = host_eval(src, params, Theta, tmp_path, name)
    # (the bar VERDICT r4 item 4 sets for the gradient: 1e-13 relative -- measured against each chain's gradient scale, which
    # is what a cancellation-free formula can promise)
    scale = np.abs(gr.numpy()).max(axis=1, keepdims=True)
    assert np.abs(g_c - gr.numpy()).max() <= 1e-13 * scale.max(), (name, np.abs(g_c - gr.numpy()).max())
    np.testing.assert_allclose(g_c, gr.numpy(), rtol=2e-13, atol=1e-13 * scale.max())
    np.testing.assert_allclose(lp_c, lp.detach().numpy(), rtol=1e-12, atol=1e-12)


def test_relu_and_clamp_follow_autograd_at_the_ties_and_hand_nan_on(tmp_path):
    """relu'(0) = 0 and clamp' = 1 AT a bound are PyTorch's conventions (a maximum / minimum rewrite splits the
    derivative 0.5 / 0.5 there -- and theta = 0 is a common start); NaN goes through relu and clamp, it is not dropped."""
    def fn(Th):
        return (2.0 * F.relu(Th) + 3.0 * torch.clamp(Th, -0.5, 0.5) + 5.0 * Th.clamp_min(-1.0) + 7.0 * Th.clamp(max=1.5)
                - 0.5 * Th * Th).sum(1)

    src, params, _ = trace.term_source(fn, D, "cd")
    ties = torch.tensor([0.0, -0.0, 0.5, -0.5, -1.0, 1.5, 0.3, -0.7, 2.0, -2.0], dtype=torch.float64)
    Theta = ties.repeat((D + len(ties) - 1) // len(ties))[:D].repeat(3, 1).clone()
    Theta[1] = Theta[1].flip(0)
    Theta[2] = torch.randn(D, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    x = Theta.clone().requires_grad_(True)
    lp = fn(x)
    (gr,) = torch.autograd.grad(lp.sum(), x)
    lp_c, g_c = host_eval(src, params, Theta, tmp_path, "ties")
    np.testing.assert_allclose(g_c, gr.numpy(), rtol=1e-13, atol=1e-13)   # (a wrong tie rule is an error of 1 .. 3.5 here)
    np.testing.assert_allclose(lp_c, lp.detach().numpy(), rtol=1e-13)
    nan = Theta.clone()
    nan[0, 4] = float("nan")
    lp_n, g_n = host_eval(src, params, nan, tmp_path, "ties_nan")
    assert np.isnan(lp_n[0]) and np.isnan(g_n[0, 4]) and np.isfinite(g_n[0, :4]).all() and np.isfinite(lp_n[1:]).all()


def test_constants_are_packed_once_and_the_source_is_plain(tmp_path):
    src, params, info = trace.term_source(lambda Th: (lam * Th * Th + lam * Th).sum(1), D)
    assert info["param_rows"] == 1 and params.shape == (D,) and torch.equal(params, lam)
    src2, params2, info2 = trace.term_source(lambda Th: (lam * Th * Th + b * Th).sum(1), D)
    assert info2["param_rows"] == 2 and torch.equal(params2, torch.cat([lam, b]))
    assert "bk_term" in src and "__device__" in src and f"P[{D} + d]" in src2
    _, none, info3 = trace.term_source(lambda Th: (-0.5 * Th * Th).sum(1), D)
    assert none is None and info3["param_rows"] == 0


@pytest.mark.parametrize("fn, needle", [
    (lambda Th: -0.5 * (Th * Th), "does not end in a per-chain value"),
    (lambda Th: (Th[:, :3] ** 2).sum(1), "getitem"),
    (lambda Th: torch.where(Th * Th, Th, -Th).sum(1), "must be a comparison"),
    (lambda Th: torch.logsumexp(Th, 1), "unsupported operation"),
    (lambda Th: (Th * Th).sum(0), "coordinate axis"),
    (lambda Th: (Th * Th).sum(), "coordinate axis"),
    (lambda Th: torch.exp((Th * Th).sum(1)), "per-chain value"),
    (lambda Th: (Th @ torch.ones(D, D, dtype=torch.float64)).sum(1), "unsupported operation"),
    (lambda Th: (Th * torch.ones(3, dtype=torch.float64)).sum(1), "does not broadcast"),
    (lambda Th: (Th ** Th).sum(1), "pow with both"),
    (lambda Th: (Th * Th).sum(1) * (Th * Th).sum(1), "product of two per-chain values"),
    (lambda Th: (Th * Th).sum(1) if Th.sum() > 0 else -(Th * Th).sum(1), "could not trace"),
])
def test_unsupported_functions_name_the_reason(fn, needle):
    with pytest.raises(trace.Unsupported, match=needle):
        trace.term_source(fn, D)


def test_torch_model_compile_keeps_autograd_when_the_source_cannot_be_built(tmp_path, monkeypatch):
    """"anything else warns and keeps autograd" also when the traced source cannot be BUILT (no hipcc on the
    box, a refused cache directory): construction must not raise."""
    from bayes_kit_amd import _lib, targets

    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(tmp_path / "cache"))

    def no_hipcc():
        raise _lib.BkHipError("no hipcc was found (test)")

    monkeypatch.setattr(targets, "_find_hipcc", no_hipcc)
    with pytest.warns(UserWarning, match="could not be built"):
        m = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), D, compile=True)
    assert m.compiled is None and "no hipcc" in m.compile_note and "bk_term" in m.traced_source
    Th = torch.randn(3, D, dtype=torch.float64)
    lp, g = m.log_density_gradient(Th)      # autograd still serves the model
    assert torch.allclose(g, -(Th * lam)) and lp.shape == (3,)


def test_torch_model_compile_flag_on_the_build_box(tmp_path, monkeypatch):
    """compile=True: a traceable function becomes a compiled target (hipcc cross-compiles here; no compute without a GPU);
    an untraceable one warns, names the node, and stays an autograd model."""
    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(tmp_path / "cache"))
    m = bk.TorchModel(lambda Th: -0.5 * (Th * Th * lam).sum(dim=1), D, compile=True)
    assert m.compiled is not None and m.compile_note is None and "bk_term" in m.traced_source
    assert m.bk_counted and hasattr(m, "bk_eval") and hasattr(m, "bk_hmc_draw") and hasattr(m, "bk_hmc_trajectory")
    assert os.path.exists(m.compiled.source_library)
    with pytest.warns(UserWarning, match="logsumexp"):
        u = bk.TorchModel(lambda Th: torch.logsumexp(Th * lam, dim=1), D, compile=True)
    assert u.compiled is None and "logsumexp" in u.compile_note and not hasattr(u, "bk_eval")
    lp, gr = u.log_density_gradient(torch.zeros((3, D), dtype=torch.float64))  # autograd still works
    assert lp.shape == (3,) and gr.shape == (3, D)
    # the namespaced gradient-only hook: a model's own `gradient` member is not what the engine calls
    w = bk.TorchModel(lambda Th: -0.5 * (Th * Th).sum(1), D, grad_fn=lambda Th: -Th)
    assert hasattr(w, "bk_gradient") and not hasattr(w, "gradient")


# ---- head-plus-sums (hierarchical) densities -> the lane-spread form (trace_lanes.py) ------------------------------------
LANES_HOST = """
#include <math.h>
#include <stdint.h>
#include "BKHIP_MATH_H_PATH"
typedef int64_t i64;
#define __device__
%s
struct HostCtx {   // one chain, the lane context's interface with plain loops (rows summed in order)
  const double* th; double* g; i64 D; int H;
  double head(int i) const { return th[i]; }
  i64 dims() const { return D; }
  template <class F> double sum(F&& f) { double s = 0.0; for (i64 d = H; d < D; ++d) s += f(th[d], d); return s; }
  void grad_head(int i, double v) { g[i] = v; }
  template <class F> void grad(F&& f) { for (i64 d = H; d < D; ++d) g[d] = f(th[d], d); }
};
extern "C" void eval_chains(const double* th, const double* P, long long C, long long D, int H, double* lp, double* g) {
  for (long long c = 0; c < C; ++c) {
    HostCtx ctx{th + c * D, g + c * D, D, H};
    lp[c] = bk_lanes_density(ctx, P);
  }
}
"""
DL = 19
yv = torch.randn(DL - 2, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
wv = torch.rand(DL - 1, generator=torch.Generator().manual_seed(4), dtype=torch.float64) + 0.5


def t_funnel(Th):
    v, x = Th[:, 0], Th[:, 1:]
    return -(v * v) / 18.0 - 0.5 * (DL - 1) * v - 0.5 * torch.exp(-v) * (x * x).sum(dim=1)


def t_funnel_inside(Th):  # the same density with the head-derived scale INSIDE the row expression, weighted rows
    v = Th[:, 0]
    r = Th[:, 1:] * torch.exp(-0.5 * v)[:, None]
    return -(v * v) / 18.0 - 0.5 * (DL - 1) * v - 0.5 * (wv * r * r).sum(1)


def t_hier(Th):
    mu, lt, x = Th[:, 0], Th[:, 1], Th[:, 2:]
    it2 = torch.exp(-2.0 * lt)
    sq = ((x - mu[:, None]) ** 2).sum(dim=1)
    sy = ((yv - x) ** 2).sum(dim=1)
    return -0.5 * it2 * sq - (DL - 2) * lt - 0.5 * sy - (mu * mu / 50.0 + 0.5 * lt * lt)


def t_logistic_scale(Th):  # rows through a sigmoid link with a shared slope (head 0) and offset (head 1), nonlinear in the sums
    a, b, x = Th[:, 0], Th[:, 1], Th[:, 2:]
    z = a.unsqueeze(1) * x + b[:, None]
    ll = F.logsigmoid(z).sum(1)
    pen = torch.log1p((x * x).sum(1))
    return ll - 0.5 * pen * pen - 0.5 * (a * a + b * b)


def t_funnel_distributions(Th):  # the funnel the way a PyTorch user writes it
    Normal = torch.distributions.Normal
    v, x = Th[:, 0], Th[:, 1:]
    return Normal(0.0, 3.0).log_prob(v) + Normal(0.0, torch.exp(0.5 * v)[:, None]).log_prob(x).sum(-1)


def t_groups(Th):  # two groups of rows with their own hyper-parameters, a third slice that overlaps neither
    mu, lt = Th[:, 0], Th[:, 1]
    ga, gb, gc = Th[:, 2:7], Th[:, 7:12], Th[:, 12:]
    la = torch.distributions.Normal(mu[:, None], torch.exp(lt)[:, None]).log_prob(ga).sum(-1)
    lb = torch.distributions.Laplace(0.0, 2.0).log_prob(gb - mu.unsqueeze(1)).sum(-1)
    lc = -0.5 * (wv[:DL - 12] * gc * gc).sum(-1) / gc.shape[1] + 0.3 * torch.tanh(gc).mean(-1) * lt
    return la + lb + lc - 0.5 * (mu * mu + lt * lt)


def t_no_heads(Th):  # no head coordinate at all: a nonlinear function of two sums over the whole state
    return -0.5 * (Th ** 2).sum(-1) - torch.log(torch.exp(Th).sum(-1)) + 0.1 * torch.tanh(Th[:, 0:].sum(1))


def t_piecewise_rows(Th):
    s, x = Th[:, 0], Th[:, 1:]
    z = x * torch.exp(-s)[:, None]
    return (torch.where(z > 0.0, -z, 2.0 * z) - F.relu(z - 0.5) ** 2 - torch.clamp(x, -0.4, 0.6) ** 2).sum(1) \
        - 0.5 * s * s - (DL - 1) * torch.maximum(s, 0.1 * s)


LANES = {"funnel_torch_distributions": (t_funnel_distributions, 1), "row_groups": (t_groups, 2), "no_heads": (t_no_heads, 0),
         "piecewise_rows": (t_piecewise_rows, 1), "funnel": (t_funnel, 1), "funnel_scale_inside_rows": (t_funnel_inside, 1), "hierarchical_normal": (t_hier, 2),
         "logistic_link_nonlinear_in_sums": (t_logistic_scale, 2)}


@pytest.mark.parametrize("name", sorted(LANES))
def test_lanes_source_value_and_gradient_match_autograd(name, tmp_path):
    from bayes_kit_amd import trace_lanes

    fn, H = LANES[name]
    src, head, params, info = trace_lanes.lanes_source(fn, DL)
    assert head == H and info["sums"] >= 1
    cpp, lib = tmp_path / f"{name}.cpp", tmp_path / f"lib{name}.so"
    cpp.write_text((LANES_HOST % src).replace("BKHIP_MATH_H_PATH", _BKHIP_MATH_H))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-shared", "-fPIC", str(cpp), "-o", str(lib)])
    h = ctypes.CDLL(str(lib))
    C = 33
    Theta = 0.6 * torch.randn((C, DL), generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    x = Theta.clone().requires_grad_(True)
    lp = fn(x)
    (gr,) = torch.autograd.grad(lp.sum(), x)
    th = np.ascontiguousarray(Theta.numpy())
    P = np.zeros(1) if params is None else np.ascontiguousarray(params.numpy())
    lp_c, g_c = np.empty(C), np.zeros((C, DL))
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    h.eval_chains(as_p(th), as_p(P), ctypes.c_longlong(C), ctypes.c_longlong(DL), ctypes.c_int(H), as_p(lp_c), as_p(g_c))
    np.testing.assert_allclose(lp_c, lp.detach().numpy(), rtol=1e-12, atol=1e-12)
    scale = np.abs(gr.numpy()).max()
    np.testing.assert_allclose(g_c, gr.numpy(), rtol=1e-11, atol=1e-12 * scale)


@pytest.mark.parametrize("fn, needle", [
    (lambda Th: (Th[:, 1:10] * Th[:, 10:]).sum(1), "ONE slice"),
    (lambda Th: (Th[:, 0] * Th[:, 1:]).sum(1), "without \\[:, None\\]"),
    (lambda Th: ((Th[:, 1:] ** 2).sum(1)[:, None] * Th[:, 1:]).sum(1), "inside another row expression"),
    (lambda Th: Th[:, 0] + Th[:, 3] + (Th[:, 2:] ** 2).sum(1), "inside the row slice"),
    (lambda Th: (Th[:, 1:25] ** 2).sum(1), "indexed as"),
    (lambda Th: (Th[:, 1::2] ** 2).sum(1), "indexed as"),
    (lambda Th: ((yv - Th[:, 0][:, None]) ** 2).sum(1) + (Th[:, 2:] ** 2).sum(1), "which rows it"),
    (lambda Th: Th[:, 0] * 2.0, "no row slice"),
])
def test_lanes_unsupported_shapes_name_the_reason(fn, needle):
    from bayes_kit_amd import trace_lanes

    with pytest.raises(trace.Unsupported, match=needle):
        trace_lanes.lanes_source(fn, DL)


def test_torch_model_compiles_a_hierarchical_density_into_the_lanes_form(tmp_path, monkeypatch):
    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(tmp_path / "cache"))
    m = bk.TorchModel(t_funnel, DL, compile=True)
    assert m.compiled is not None and m.compiled_form == "lanes" and "bk_lanes_density" in m.traced_source
    for hook in ("bk_eval", "bk_leapfrog_step", "bk_hmc_proposal", "bk_dr_proposal"):
        assert hasattr(m, hook), hook
    assert m.bk_dr_proposal_supported() and not hasattr(m, "bk_hmc_draw")
    e = bk.TorchModel(lambda Th: -0.5 * (Th * Th).sum(1), DL, compile=True)
    assert e.compiled_form == "elementwise"
    with pytest.warns(UserWarning, match="as head coordinates plus sums over rows"):
        u = bk.TorchModel(lambda Th: torch.logsumexp(Th, dim=1), DL, compile=True)
    assert u.compiled is None


# ---- coordinates coupled through shifted slices -> the per-chain form (trace_chain.py) -----------------------------------
CHAIN_HOST = """
#include <math.h>
#include <stdint.h>
#include "BKHIP_MATH_H_PATH"
typedef int64_t i64;
#define __device__
struct BkTheta { const double* p; double operator[](i64 d) const { return p[d]; } void fence() const {} };
#define __forceinline__ inline
#define BK_CHAIN_PACE(i)
#define BK_CHAIN_FORGET(p)
struct BkGrad { double* p; bool wanted() const { return p != nullptr; } void set(i64 d, double v) const { p[d] = v; } };
%s
extern "C" void eval_chains(const double* th, const double* P, long long C, long long D, double* lp, double* g) {
  for (long long c = 0; c < C; ++c) { BkTheta t{th + c * D}; BkGrad gr{g ? g + c * D : nullptr}; lp[c] = bk_chain(t, gr, D, P); }
}
"""
DC = 24
yc = torch.randn(DC - 2, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
yc1 = torch.cat([yc, yc[:1]])


def c_ar1(Th):  # AR(1) latent path with a learned correlation and scale, Gaussian observations, stationary first state
    phi, ls, x = torch.tanh(Th[:, 0]), Th[:, 1], Th[:, 2:]
    inn = x[:, 1:] - phi[:, None] * x[:, :-1]
    return -0.5 * (inn * inn).sum(-1) * torch.exp(-2 * ls) - (DC - 3) * ls - 0.5 * x[:, 0] ** 2 * (1 - phi * phi) * torch.exp(-2 * ls) \
        - 0.5 * ((yc - x) ** 2).sum(-1) - 0.5 * Th[:, 0] ** 2 - 0.5 * ls * ls


def c_rw2(Th):  # second-order random-walk (smoothness) prior on the whole state
    d2 = torch.diff(torch.diff(Th, dim=1), dim=1)
    return -0.5 * (d2 ** 2).sum(1) * 4.0 - 0.05 * (Th ** 2).sum(1)


def c_stochastic_volatility(Th):  # h_t AR(1) around mu, y_t ~ N(0, exp(h_t))
    mu, h = Th[:, 0], Th[:, 1:]
    return -0.5 * ((h[:, 1:] - mu[:, None] - 0.9 * (h[:, :-1] - mu[:, None])) ** 2).sum(-1) / 0.04 \
        - 0.5 * (h + yc1 * yc1 * torch.exp(-h)).sum(-1) - 0.5 * mu * mu


def c_mixed_offsets(Th):  # several offsets, slices of computed vectors, scalars taken from the middle and the end, piecewise
    inc = torch.diff(Th, dim=1)
    return -torch.sqrt(1e-2 + inc ** 2).sum(-1) - 0.5 * ((Th[:, 3:] - Th[:, :-3]) ** 2).mean(-1) - 0.1 * Th[:, 5] ** 2 * Th[:, -1] \
        - 0.5 * (Th * Th).sum(-1) - torch.relu(inc[:, 2:] * inc[:, :-2]).sum(1) - (inc * inc)[:, 4] * torch.tanh(Th[:, 0])


CHAINS = {"ar1": c_ar1, "second_order_random_walk": c_rw2, "stochastic_volatility": c_stochastic_volatility,
          "mixed_offsets": c_mixed_offsets}


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_chain_source_value_and_gradient_match_autograd(name, tmp_path):
    from bayes_kit_amd import trace_chain

    fn = CHAINS[name]
    src, params, info = trace_chain.chain_source(fn, DC)
    assert info["sums"] >= 1 and info["gradient_blocks"] >= 2 and "#pragma unroll" in src
    cpp, lib = tmp_path / f"{name}.cpp", tmp_path / f"lib{name}.so"
    cpp.write_text((CHAIN_HOST % src).replace("BKHIP_MATH_H_PATH", _BKHIP_MATH_H))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-shared", "-fPIC", str(cpp), "-o", str(lib)])
    h = ctypes.CDLL(str(lib))
    C = 33
    Theta = 0.6 * torch.randn((C, DC), generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    x = Theta.clone().requires_grad_(True)
    lp = fn(x)
    (gr,) = torch.autograd.grad(lp.sum(), x)
    th = np.ascontiguousarray(Theta.numpy())
    P = np.zeros(1) if params is None else np.ascontiguousarray(params.numpy())
    lp_c, g_c = np.empty(C), np.zeros((C, DC))
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    h.eval_chains(as_p(th), as_p(P), ctypes.c_longlong(C), ctypes.c_longlong(DC), as_p(lp_c), as_p(g_c))
    np.testing.assert_allclose(lp_c, lp.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g_c, gr.numpy(), rtol=1e-11, atol=1e-12 * np.abs(gr.numpy()).max())
    lp_only = np.empty(C)   # (no gradient wanted: the log density alone)
    h.eval_chains(as_p(th), as_p(P), ctypes.c_longlong(C), ctypes.c_longlong(DC), as_p(lp_only), None)
    np.testing.assert_array_equal(lp_only, lp_c)


@pytest.mark.parametrize("fn, needle", [
    (lambda Th: (Th[:, 1:] * Th[:, :-2]).sum(1), "different lengths"),
    (lambda Th: (Th[:, ::2] ** 2).sum(1), "strided"),
    (lambda Th: ((Th ** 2).sum(1)[:, None] * Th).sum(1), "inside another vector expression"),
    (lambda Th: (Th @ torch.ones(DC, DC, dtype=torch.float64)).sum(1), "unsupported operation"),
    (lambda Th: torch.cumsum(Th, 1).sum(1), "unsupported operation"),
    (lambda Th: (Th[:, 1:] * Th[:, 0]).sum(1), "without \\[:, None\\]"),
])
def test_chain_unsupported_shapes_name_the_reason(fn, needle):
    from bayes_kit_amd import trace_chain

    with pytest.raises(trace.Unsupported, match=needle):
        trace_chain.chain_source(fn, DC)


def test_torch_model_compiles_a_coupled_density_into_the_per_chain_form(tmp_path, monkeypatch):
    monkeypatch.setenv("BK_SOURCE_TARGET_DIR", str(tmp_path / "cache"))
    m = bk.TorchModel(c_ar1, DC, compile=True)
    assert m.compiled is not None and m.compiled_form == "chain" and "bk_chain" in m.traced_source, m.compile_note
    for hook in ("bk_eval", "bk_leapfrog_step", "bk_leapfrog_trajectory"):
        assert hasattr(m, hook), hook
    assert m.bk_counted and not hasattr(m, "bk_dr_proposal")
    with pytest.warns(UserWarning, match="as sums over shifted slices"):
        u = bk.TorchModel(lambda Th: torch.cumsum(Th, 1).sum(1), DC, compile=True)
    assert u.compiled is None


# ---- randomised expressions: every derivative rule of both tracers against autograd --------------------------------------
def _random_expr(rng, depth, leaves):
    """A random elementwise PyTorch expression (as a Python closure) over the supported operations, kept in a numerically tame
    range (arguments of log / sqrt / pow are made positive, exponents bounded)."""
    if depth == 0 or rng.random() < 0.15:
        return leaves[int(rng.integers(0, len(leaves)))]
    kind = rng.random()
    a = _random_expr(rng, depth - 1, leaves)
    if kind < 0.45:
        b = _random_expr(rng, depth - 1, leaves)
        op = int(rng.integers(0, 4))
        if op == 0:
            return lambda x: a(x) + b(x)
        if op == 1:
            return lambda x: a(x) - b(x)
        if op == 2:
            return lambda x: a(x) * b(x)
        return lambda x: a(x) / (2.0 + torch.square(b(x)))
    c = float(rng.uniform(0.3, 1.7))
    u = int(rng.integers(0, 20))
    table = [
        lambda x: torch.log(torch.cosh(a(x))), lambda x: torch.sinh(torch.tanh(a(x)) * c), lambda x: torch.atan(a(x) * c),
        lambda x: torch.erf(a(x)), lambda x: torch.where(a(x) > 0.1 * c, torch.sin(a(x)), 0.5 * a(x)),
        lambda x: torch.clamp(a(x), -c, c) * torch.minimum(a(x), 0.5 * a(x) + 0.1),
        lambda x: torch.exp(-torch.square(a(x)) * c), lambda x: torch.log(1.5 + torch.square(a(x))), lambda x: torch.log1p(torch.square(a(x)) * c),
        lambda x: torch.expm1(-torch.abs(a(x))), lambda x: torch.sigmoid(a(x) * c), lambda x: F.logsigmoid(a(x)), lambda x: F.softplus(a(x) * c),
        lambda x: torch.tanh(a(x)), lambda x: torch.sqrt(1.0 + torch.square(a(x))), lambda x: torch.square(a(x)), lambda x: torch.sin(a(x) * c),
        lambda x: torch.cos(a(x)), lambda x: (1.0 + torch.square(a(x))) ** c, lambda x: -a(x) * c + 2.0 ** (0.3 * torch.tanh(a(x))),
    ]
    return table[u]


def test_randomised_separable_densities_against_autograd(tmp_path):
    """40 random elementwise expression trees (depth <= 4) over every supported operation, each summed over the coordinates:
    value and hand-differentiated derivative of the generated bk_term against the function and autograd."""
    rng = np.random.default_rng(20250)
    pw = torch.rand(D, generator=torch.Generator().manual_seed(9), dtype=torch.float64) + 0.5
    leaves = [lambda x: x, lambda x: x * pw, lambda x: x - a, lambda x: 0.5 * x + 0.25]
    fns, srcs = [], []
    for k in range(40):
        e = _random_expr(rng, 4, leaves)
        fn = (lambda e_: (lambda Th: e_(Th).sum(dim=1)))(e)
        src, params, _ = trace.term_source(fn, D)
        srcs.append((src.replace("bk_term(", f"bk_term_{k}("), params))
        fns.append(fn)
    body = "\n".join(s for s, _ in srcs)
    calls = "\n".join(f"    case {k}: bk_term_{k}(th[i], d[i], P, term[i], grad[i]); break;" for k in range(len(fns)))
    cpp = ("#include <math.h>\n#include <stdint.h>\n#include \"BKHIP_MATH_H_PATH\"\ntypedef int64_t i64;\n#define __device__\n#define __forceinline__ inline\n" + body +
           "\nextern \"C\" void eval_k(int k, const double* th, const long long* d, const double* P, long long n, double* term, double* grad) {\n"
           "  for (long long i = 0; i < n; ++i) switch (k) {\n" + calls + "\n  }\n}\n")
    (tmp_path / "rnd.cpp").write_text(cpp.replace("BKHIP_MATH_H_PATH", _BKHIP_MATH_H))
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", str(tmp_path / "rnd.cpp"), "-o", str(tmp_path / "librnd.so")])
    h = ctypes.CDLL(str(tmp_path / "librnd.so"))
    C = 16
    Theta = 0.8 * torch.randn((C, D), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    th = np.ascontiguousarray(Theta.numpy().reshape(-1))
    dd = np.ascontiguousarray(np.tile(np.arange(D, dtype=np.int64), C))
    as_p = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for k, fn in enumerate(fns):
        x = Theta.clone().requires_grad_(True)
        lp = fn(x)
        (gr,) = torch.autograd.grad(lp.sum(), x)
        P = np.zeros(1) if srcs[k][1] is None else np.ascontiguousarray(srcs[k][1].numpy())
        term, grad = np.empty(C * D), np.empty(C * D)
        h.eval_k(ctypes.c_int(k), as_p(th), as_p(dd), as_p(P), ctypes.c_longlong(C * D), as_p(term), as_p(grad))
        scale = max(1.0, float(np.abs(gr.numpy()).max()))
        np.testing.assert_allclose(grad.reshape(C, D), gr.numpy(), rtol=1e-10, atol=1e-12 * scale, err_msg=f"expression {k}")
        np.testing.assert_allclose(term.reshape(C, D).sum(axis=1), lp.detach().numpy(), rtol=1e-11, atol=1e-11, err_msg=f"expression {k}")


def test_randomised_hierarchical_densities_against_autograd(tmp_path):
    """25 random head-plus-sums densities: random row expressions that use two head coordinates through [:, None], two or three
    sums, a random scalar expression of heads and sums at the end -- the symbolic derivatives of trace_lanes.py (incl. the extra
    sums the head gradients need) against autograd."""
    from bayes_kit_amd import trace_lanes

    rng = np.random.default_rng(77)
    n = DL - 2
    pw = torch.rand(n, generator=torch.Generator().manual_seed(8), dtype=torch.float64) + 0.5
    fns, srcs = [], []
    for k in range(25):
        def make():
            row_leaves = lambda h0, h1: [lambda x: x, lambda x: x * pw, lambda x: x - h0, lambda x: x * torch.exp(-0.3 * h1), lambda x: 0.5 * x + h0 * 0.1]  # noqa: E731
            seeds = [int(s) for s in rng.integers(0, 2 ** 31, size=3)]
            nsum = int(rng.integers(2, 4))
            mode = int(rng.integers(0, 3))

            def fn(Th):
                h0, h1, x = Th[:, 0], Th[:, 1], Th[:, 2:]
                leaves = row_leaves(h0[:, None], h1[:, None])
                sums = [_random_expr(np.random.default_rng(seeds[i]), 3, leaves)(x).sum(dim=1) for i in range(nsum)]
                s = sums[0] - 0.5 * (h0 * h0 + h1 * h1)
                if mode == 0:
                    s = s + torch.tanh(sums[1] * 0.1) * torch.exp(-0.2 * h0)
                elif mode == 1:
                    s = s - torch.log1p(torch.square(sums[1])) * torch.sigmoid(h1)
                else:
                    s = s + sums[1] / (2.0 + torch.square(h0 - h1))
                if nsum == 3:
                    s = s - 0.3 * torch.sqrt(1.0 + torch.square(sums[2]))
                return s
            return fn
        fn = make()
        src, head, params, info = trace_lanes.lanes_source(fn, DL)
        assert head == 2
        srcs.append((src.replace("bk_lanes_density(", f"bk_lanes_density_{k}("), params))
        fns.append(fn)
    body = "\n".join(s for s, _ in srcs)
    calls = "\n".join(f"      case {k}: lp[c] = bk_lanes_density_{k}(ctx, P); break;" for k in range(len(fns)))
    host = LANES_HOST.split("extern \"C\"")[0] % body
    cpp = host + ("extern \"C\" void eval_k(int k, const double* th, const double* P, long long C, long long D, int H, double* lp, double* g) {\n"
                  "  for (long long c = 0; c < C; ++c) {\n    HostCtx ctx{th + c * D, g + c * D, D, H};\n    switch (k) {\n" + calls +
                  "\n    }\n  }\n}\n")
    (tmp_path / "rndl.cpp").write_text(cpp.replace("BKHIP_MATH_H_PATH", _BKHIP_MATH_H))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-shared", "-fPIC", str(tmp_path / "rndl.cpp"), "-o",
                           str(tmp_path / "librndl.so")])
    h = ctypes.CDLL(str(tmp_path / "librndl.so"))
    C = 12
    Theta = 0.6 * torch.randn((C, DL), generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    th = np.ascontiguousarray(Theta.numpy())
    as_p = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for k, fn in enumerate(fns):
        x = Theta.clone().requires_grad_(True)
        lp = fn(x)
        (gr,) = torch.autograd.grad(lp.sum(), x)
        P = np.zeros(1) if srcs[k][1] is None else np.ascontiguousarray(srcs[k][1].numpy())
        lp_c, g_c = np.empty(C), np.zeros((C, DL))
        h.eval_k(ctypes.c_int(k), as_p(th), as_p(P), ctypes.c_longlong(C), ctypes.c_longlong(DL), ctypes.c_int(2), as_p(lp_c), as_p(g_c))
        scale = max(1.0, float(np.abs(gr.numpy()).max()))
        np.testing.assert_allclose(lp_c, lp.detach().numpy(), rtol=1e-11, atol=1e-11, err_msg=f"density {k}")
        np.testing.assert_allclose(g_c, gr.numpy(), rtol=1e-9, atol=1e-11 * scale, err_msg=f"density {k}")
