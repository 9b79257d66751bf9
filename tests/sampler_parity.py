"""Parity bodies shared by the CPU (fake ops) and GPU (HIP) sampler tests."""
import numpy as np
import torch

import bayes_kit_amd as bk
from tests.helpers import case_metric, case_seed, host_proposal, load_case, oracle_model

# Tolerances (DESIGN.md "Parity"): theta bit-exact for elementwise-gradient targets; logp and
# energies are reductions summed in a different order than BLAS ddot -> rel 1e-12.
# Funnel (exp + a reduction inside the gradient): SURVEY 8c's bar is rel 1e-9 -- held for the
# first 20 draws (measured on MI355X, tools/funnel_parity_report.py -> profiles/r2_funnel_parity.md:
# max abs error 1.8e-12, max rel 1.7e-10 in the first 20 draws of both funnel fixtures; the absolute
# floor 5e-11 = the relative bar at |x| = 0.05 covers coordinates that pass through zero).
# The Hamiltonian flow on the funnel is chaotic and the amplification is not smooth: one 160-step
# trajectory through the funnel's neck multiplies a last-bit difference (summation order, exp) by
# 10-100 at once (fixture funnel101: 9.5e-13 at draw 19 -> 1.3e-9 at draw 21), after which the error
# drifts up by ~1.1x per draw.  From draw 20 on the bar is therefore rel 1e-7 / abs 1e-8, widening by
# 1.1x per draw.  A single flipped accept / retry decision would be an O(1) jump, and the bit
# generator's final state, which pins every decision's RNG consumption, must match exactly.
LOGP_RTOL = 1e-12
FUNNEL_EXACT_DRAWS = 20
FUNNEL_RTOL0, FUNNEL_ATOL0 = 1e-9, 5e-11
FUNNEL_LATE_RTOL0, FUNNEL_LATE_ATOL0, FUNNEL_GROWTH = 1e-7, 1e-8, 1.1


def funnel_tol(n):
    if n < FUNNEL_EXACT_DRAWS:
        return dict(rtol=FUNNEL_RTOL0, atol=FUNNEL_ATOL0)
    g = FUNNEL_GROWTH ** (n - FUNNEL_EXACT_DRAWS)
    return dict(rtol=FUNNEL_LATE_RTOL0 * g, atol=FUNNEL_LATE_ATOL0 * g)


def report_funnel_error(name, n, got, want):
    """One line per draw in the test log (pytest -s / on failure): the measured parity error."""
    err = np.abs(got - want)
    print(f"funnel parity {name} draw {n:3d}: max abs err {err.max():.2e}, max err/(atol+rtol|x|) "
          f"{(err / (funnel_tol(n)['atol'] + funnel_tol(n)['rtol'] * np.abs(want))).max():.3f}")


def product_model(spec, ops):
    kind = spec["kind"]
    if kind == "std_normal":
        return bk.IsoGaussian(1, ops=ops)
    if kind == "iso_gaussian":
        return bk.IsoGaussian(spec["D"], ops=ops)
    if kind == "diag_gaussian":
        return bk.DiagGaussian(np.logspace(spec["log10_lo"], spec["log10_hi"], spec["D"]), ops=ops)
    if kind == "funnel":
        return bk.Funnel(spec["D"], ops=ops)
    raise KeyError(kind)


def build_sampler(case, model, ops, seed, chains=None, chain_id0=0, **extra):
    D = model.dims()
    init = None if case.get("init") is None else np.asarray(case["init"], dtype=np.float64)
    kw = dict(init=init, seed=seed, ops=ops, **extra)
    if chains is not None:
        kw.update(chains=chains, chain_id0=chain_id0)
    alg = case["alg"]
    if alg == "hmc":
        s = bk.HMCDiag(model, case["stepsize"], case["steps"], **kw)
    elif alg == "mala":
        s = bk.MALA(model, case["epsilon"], **kw)
    elif alg == "drghmc":
        s = bk.DrGhmcDiag(model, case["max_proposals"], case["leapfrog_step_sizes"],
                          case["leapfrog_step_counts"], case["damping"],
                          prob_retry=case.get("prob_retry", True), **kw)
    elif alg in ("metropolis", "mh"):
        kw.pop("path", None)
        if chains is not None:  # every chain in one sampler: batched callbacks on device streams
            proposal_fn, transition_lp_fn = device_proposal(case["proposal"], chains, chain_id0, ops)
        else:
            proposal_fn, transition_lp_fn = host_proposal(case["proposal"], extra_chain(seed, case))
        if alg == "metropolis":
            s = bk.Metropolis(model, proposal_fn, **kw)
        else:
            s = bk.MetropolisHastings(model, proposal_fn, transition_lp_fn, **kw)
    else:
        raise KeyError(alg)
    metric = case_metric(case, D)
    if metric is not None:
        s._metric = metric  # same hook the golden generator uses on the reference
    return s


def extra_chain(seed, case):
    """Chain index a single-chain golden seed belongs to (the proposal stream is keyed by it)."""
    if "pcg_seed" in case:
        return int(seed) - case["pcg_seed"]
    return int(seed.state["state"]["key"][1])


def device_proposal(spec, chains, chain_id0, ops):
    """The many-chain form of tests.helpers.host_proposal: one call proposes for every chain, the
    normals coming from ChainRng streams keyed like the single-chain proposals' generators."""
    from bayes_kit_amd.metropolis import ChainRng

    prng = ChainRng(spec["seed"], chains, chain_id0, ops=ops)
    a, scale = spec.get("a", 1.0), spec["scale"]

    def proposal_fn(Theta):
        return prng.normal(a * Theta, scale)

    def transition_lp_fn(To, From):
        r = (To - a * From) / scale
        return -0.5 * (r * r).sum(dim=1)

    return proposal_fn, transition_lp_fn


def check_many_chain(name, ops, model_factory=None, **extra):
    """All C chains of a golden case in ONE many-chain sampler on a built-in device target (model_factory(spec, ops):
    another provider of the same density, e.g. one compiled from source)."""
    case, z = load_case(name)
    N, C, D = z["draws"].shape
    model = (model_factory or product_model)(case["model"], ops)
    s = build_sampler(case, model, ops, case["seed"], chains=C, **extra)
    exact = case["model"]["kind"] != "funnel"
    th0 = s._theta.cpu().numpy()
    np.testing.assert_array_equal(th0, z["theta0"])
    for n in range(N):
        th, lp = s.sample()
        th, lp = th.cpu().numpy(), lp.cpu().numpy()
        if exact:
            assert np.array_equal(th, z["draws"][n]), (name, n, np.abs(th - z["draws"][n]).max())
        else:
            report_funnel_error(name, n, th, z["draws"][n])
            np.testing.assert_allclose(th, z["draws"][n], err_msg=f"{name} draw {n}", **funnel_tol(n))
        tol = dict(rtol=LOGP_RTOL, atol=1e-12) if exact else funnel_tol(n)
        np.testing.assert_allclose(lp, z["logp"][n], err_msg=f"{name} draw {n}", **tol)
    # integer side: the per-chain streams end exactly where numpy's did
    np.testing.assert_array_equal(s.rng_state().T, z["rng_state"])
    if case["alg"] == "drghmc":
        rho = s._rho.cpu().numpy()
        if exact:
            np.testing.assert_array_equal(rho, z["rho_final"])
        else:
            np.testing.assert_allclose(rho, z["rho_final"], **funnel_tol(N))
    return s


def check_single_chain_host_model(name, ops, chains=None):
    """Reference-style NumPy model + reference-style seed through the drop-in classes."""
    case, z = load_case(name)
    N, C, D = z["draws"].shape
    exact = case["model"]["kind"] != "funnel"
    for c in (range(C) if chains is None else chains):
        s = build_sampler(case, oracle_model(case["model"]), ops, case_seed(case, c))
        np.testing.assert_array_equal(np.asarray(s._theta, dtype=np.float64), z["theta0"][c])
        for n in range(N):
            th, lp = s.sample()
            assert isinstance(th, np.ndarray) and th.shape == (D,)
            if exact:
                assert np.array_equal(th, z["draws"][n, c]), (name, c, n)
            else:
                np.testing.assert_allclose(th, z["draws"][n, c], **funnel_tol(n))
            tol = dict(rtol=LOGP_RTOL, atol=1e-12) if exact else funnel_tol(n)
            np.testing.assert_allclose(lp, z["logp"][n, c], **tol)
        got = s.rng_state()[:, 0]
        want = z["rng_state"][c]
        np.testing.assert_array_equal(got[: len(want)], want)


def check_checkpoint_resume(ops, make):
    """5 draws, checkpoint, 5 more == fresh sampler restored from the checkpoint, 5 draws."""
    import io

    import torch

    a = make()
    for _ in range(5):
        a.sample()
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    more = [a.sample() for _ in range(5)]
    b = make()
    buf.seek(0)
    b.load_state_dict(torch.load(buf, weights_only=False))
    for th_a, lp_a in more:
        th_b, lp_b = b.sample()
        assert np.array_equal(np.asarray(th_a.cpu()), np.asarray(th_b.cpu()))
        assert np.array_equal(np.asarray(lp_a.cpu()), np.asarray(lp_b.cpu()))
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())


def check_dense_metric_hmc(ops, C=40, D=24, draws=6, rtol=1e-10):
    """HMC with a dense mass matrix vs the oracle's HMCDense (no reference counterpart:
    parity unpinned; tolerance because M @ v is summed in a different order)."""
    from oracle import models as om
    from oracle import samplers as osamp

    rng = np.random.default_rng(42)
    lam = np.logspace(0, 1, D)
    A = rng.normal(size=(D, D)) * 0.05
    M = np.diag(1.0 / lam) + A @ A.T / lam.max()  # near the posterior covariance: a useful metric
    s = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 7, chains=C, seed=31, metric_dense=M, ops=ops)
    outs = []
    for _ in range(draws):
        th, lp = s.sample()
        outs.append((np.asarray(th.cpu()), np.asarray(lp.cpu())))
    for c in range(0, C, max(1, C // 8)):
        o = osamp.HMCDense(om.DiagGaussian(lam), 0.05, 7, M, seed=np.random.Philox(key=[31, c]))
        for n in range(draws):
            oth, olp = o.sample()
            np.testing.assert_allclose(outs[n][0][c], oth, rtol=rtol, atol=1e-13)
            np.testing.assert_allclose(outs[n][1][c], olp, rtol=rtol, atol=1e-12)
    assert 0.3 < s.accept_rate() <= 1.0
    # M = I: the dense path IS the reference path (multiplying by 1 and adding zeros is exact)
    a = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 7, chains=C, seed=31, metric_dense=np.eye(D), ops=ops)
    b = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.05, 7, chains=C, seed=31, path="step", ops=ops)
    for _ in range(3):
        ta, la = a.sample()
        tb, lb = b.sample()
        assert np.array_equal(np.asarray(ta.cpu()), np.asarray(tb.cpu()))
        np.testing.assert_allclose(np.asarray(la.cpu()), np.asarray(lb.cpu()), rtol=1e-12)


def check_smc_binomial(ops, M, N, kernel, mean_atol, var_atol, seed=11):
    """Likelihood-tempered SMC on the conjugate beta-binomial model of the reference's
    test/test_tempered_smc.py:8-30 (logit scale, posterior Beta(alpha+x, beta+N-x))."""
    import math

    import torch

    alpha, beta, x, Nobs = 2.0, 3.0, 5.0, 15.0
    logB = math.lgamma(alpha) + math.lgamma(beta) - math.lgamma(alpha + beta)
    logC = math.lgamma(Nobs + 1) - math.lgamma(x + 1) - math.lgamma(Nobs - x + 1)

    def log_prior(Th):
        lp, l1p = torch.nn.functional.logsigmoid(Th[:, 0]), torch.nn.functional.logsigmoid(-Th[:, 0])
        return (alpha - 1) * lp + (beta - 1) * l1p - logB + lp + l1p  # beta prior + logit Jacobian

    def log_lik(Th):
        lp, l1p = torch.nn.functional.logsigmoid(Th[:, 0]), torch.nn.functional.logsigmoid(-Th[:, 0])
        return x * lp + (Nobs - x) * l1p + logC

    model = bk.TorchPriorLikelihoodModel(log_prior, log_lik, 1)
    rng = np.random.default_rng(seed)
    p0 = rng.beta(alpha, beta, size=M)
    init = np.log(p0 / (1 - p0)).reshape(M, 1)  # model.initial_state: logit of a prior draw
    smc = bk.TemperedLikelihoodSMC(model, M, N, init, kernel, seed=seed, ops=ops)
    assert smc.time(3) == 3 / N and smc.D == 1
    smc.run()
    th = np.asarray(smc.thetas.cpu())
    assert th.shape == (M, 1)
    draws = 1.0 / (1.0 + np.exp(-th[:, 0]))
    a, b = alpha + x, beta + Nobs - x
    np.testing.assert_allclose(draws.mean(), a / (a + b), atol=mean_atol)
    np.testing.assert_allclose(draws.var(ddof=1), a * b / ((a + b) ** 2 * (a + b + 1)), atol=var_atol)
    return smc


def check_logistic_target(ops, N=700, D=24, C=50):
    """LogisticRegression (two MFMA GEMMs + one elementwise pass) vs the oracle's NumPy model,
    then HMC on it vs the oracle sampler chain by chain (no reference counterpart: tolerance)."""
    import torch
    from oracle import models as om
    from oracle import samplers as osamp

    rng = np.random.default_rng(5)
    X = rng.normal(size=(N, D)) / np.sqrt(D)
    tstar = rng.normal(size=D)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ tstar))).astype(np.float64)
    model = bk.LogisticRegression(X, y, prior_scale=2.0, ops=ops)
    omodel = om.LogisticRegression(X, y, prior_scale=2.0)
    Th = rng.normal(size=(C, D))
    th_dc = torch.from_numpy(np.ascontiguousarray(Th.T)).to(ops.device)
    lp, g = model.log_density_gradient(th_dc.t())
    ll = model.log_likelihood(th_dc.t())
    lpt, gt = model.log_density_gradient_tempered(th_dc.t(), 0.3)
    for c in range(0, C, 7):
        olp, og = omodel.log_density_gradient(Th[c])
        np.testing.assert_allclose(lp[c].item(), olp, rtol=1e-11)
        np.testing.assert_allclose(np.asarray(g[c].cpu()), og, rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(ll[c].item(), omodel.log_likelihood(Th[c]), rtol=1e-11)
        np.testing.assert_allclose(lpt[c].item(), 0.3 * omodel.log_likelihood(Th[c]) + omodel.log_prior(Th[c]),
                                   rtol=1e-11)
    s = bk.HMCDiag(model, 0.05, 6, chains=C, seed=13, ops=ops)
    outs = [s.sample() for _ in range(4)]
    for c in range(0, C, 11):
        o = osamp.HMCDiag(omodel, 0.05, 6, seed=np.random.Philox(key=[13, c]))
        for th, lp in outs:
            oth, olp = o.sample()
            np.testing.assert_allclose(np.asarray(th[c].cpu()), oth, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(lp[c].item(), olp, rtol=1e-9)
    return model, tstar


def check_attached_diagnostics(ops, C=40, D=7, draws=12, **kw):
    """DrGhmcDiag.attach(): moments and recorder fed from inside the draw (device-side draw counter, part of the
    captured launch sequence) against update() / record() called after every sample(): same moments, same series."""
    import torch

    args = (3, [0.5, 0.2, 0.08], [2, 3, 5], 0.4)
    a = bk.DrGhmcDiag(bk.Funnel(D, ops=ops), *args, chains=C, seed=11, ops=ops, **kw)
    b = bk.DrGhmcDiag(bk.Funnel(D, ops=ops), *args, chains=C, seed=11, ops=ops, **kw)
    ma, mb = bk.RunningMoments(D, C, ops=ops), bk.RunningMoments(D, C, ops=ops)
    ra, rb = bk.DrawRecorder([0, D - 1], draws, C, ops=ops), bk.DrawRecorder([0, D - 1], draws, C, ops=ops)
    for _ in range(3):  # attach after a few draws: the offsets between the sampler's draw count and n matter
        a.sample()
        b.sample()
    b.attach(moments=mb, recorder=rb)
    for n in range(draws):
        th, lp = a.sample()
        ma.update(th)
        ra.record(th, lp)
        if n % 3 == 0:
            tb, lb = b.sample()
            assert torch.equal(th, tb) and torch.equal(lp, lb)
        else:
            b.advance()
        assert (mb.n, rb.n) == (ma.n, ra.n) == (n + 1, n + 1)
    assert torch.equal(ma.mean, mb.mean) and torch.equal(ma.m2, mb.m2)
    assert torch.equal(ra.series, rb.series)
    np.testing.assert_array_equal(ma.rhat(), mb.rhat())
    np.testing.assert_array_equal(a.rng_state(), b.rng_state())
    import pytest

    with pytest.raises(IndexError):
        b.advance()  # the recorder is full
    return b


def check_recorder_and_moments_edges(ops, C=12, D=5):
    """tracked coordinates are validated against D (negative = from the end, out of range raises, nothing is
    read unchecked); a square draw is told apart by strides or by layout=; a draw with padded rows feeds the Welford
    update without a staging copy; attachments survive load_state_dict() of the sampler in either order."""
    import pytest
    import torch

    dev = ops.device
    g = torch.Generator().manual_seed(5)
    dc = torch.randn((D, C), generator=g, dtype=torch.float64).to(dev)
    lp = torch.randn(C, generator=g, dtype=torch.float64).to(dev)
    # negative and out-of-range tracked coordinates
    r = bk.DrawRecorder([0, -1, -D], 4, C, ops=ops)
    r.record(dc.t(), lp)          # sample()'s (C, D) view
    r.record(dc, lp)              # the [D, C] buffer
    r.record(dc.t().contiguous(), lp)  # a row-major (C, D) copy: the strided fallback
    for row in range(3):
        assert torch.equal(r.series[0, row], dc[0]) and torch.equal(r.series[1, row], dc[D - 1])
        assert torch.equal(r.series[2, row], dc[0]) and torch.equal(r.series[3, row], lp)
    for bad in ([D], [-D - 1], [0, 10 ** 6]):
        with pytest.raises(IndexError):
            bk.DrawRecorder(bad, 2, C, ops=ops).record(dc.t(), lp)
    # a square draw: strides decide, an ambiguous one needs layout=
    sq = torch.randn((C, C), generator=g, dtype=torch.float64).to(dev)   # [D = C, C] buffer
    rs = bk.DrawRecorder([1], 3, C, ops=ops)
    rs.record(sq.t(), lp)                        # the (C, D) view of the buffer: stride(0) == 1
    rs.record(sq, lp)                            # the buffer itself
    rs.record(sq.t().contiguous(), lp, layout="cd")  # a contiguous (C, D) copy says what it is
    for row in range(3):
        assert torch.equal(rs.series[0, row], sq[1]), row
    if C > 1:
        one = torch.ones((1, 1), dtype=torch.float64, device=dev).expand(C, C)  # strides (0, 0): neither layout
        with pytest.raises(ValueError):
            bk.DrawRecorder([1], 1, C, ops=ops).record(one, lp)
    # Welford update straight from a draw whose rows are padded (no .contiguous() staging copy)
    padded = torch.zeros((D, C + 6), dtype=torch.float64, device=dev)[:, :C]
    ma, mb = bk.RunningMoments(D, C, ops=ops), bk.RunningMoments(D, C, ops=ops)
    for k in range(3):
        x = torch.randn((D, C), generator=g, dtype=torch.float64).to(dev)
        padded.copy_(x)
        ma.update(x)
        mb.update(padded.t() if k % 2 else padded)
    assert torch.equal(ma.mean, mb.mean) and torch.equal(ma.m2, mb.m2)
    # attach() offsets after restoring the sampler (the diagnostics keep counting from where THEY are)
    args = (2, [0.5, 0.2], [2, 3], 0.4)
    mk = lambda: bk.DrGhmcDiag(bk.Funnel(D, ops=ops), *args, chains=C, seed=3, ops=ops)  # noqa: E731
    a, b = mk(), mk()
    m_a, m_b = bk.RunningMoments(D, C, ops=ops), bk.RunningMoments(D, C, ops=ops)
    r_b = bk.DrawRecorder([0], 8, C, ops=ops)
    for _ in range(2):
        a.sample()
        b.sample()
    sd = b.state_dict()           # the sampler at draw 2
    b.attach(moments=m_b, recorder=r_b)
    for _ in range(3):
        b.advance()               # draws 3..5 seen by the diagnostics (n = 3)
    b.load_state_dict(sd)         # back to draw 2; the diagnostics are NOT rewound: they go on at n = 3, 4, ...
    for _ in range(2):
        b.advance()
    assert m_b.n == 5 and r_b.n == 5
    ref = mk()
    for _ in range(2):
        ref.sample()
    seen = []
    for _ in range(3):
        seen.append(ref.sample()[0].t().clone())
    ref.load_state_dict(sd)
    for _ in range(2):
        seen.append(ref.sample()[0].t().clone())
    for x in seen:
        m_a.update(x)
    assert torch.equal(m_a.mean, m_b.mean) and torch.equal(m_a.m2, m_b.m2)
    for row, x in enumerate(seen):
        assert torch.equal(r_b.series[0, row], x[0]), row


def check_adaptive_smc_ladder(ops, M=600, D=3, n_obs=4000, seed=5):
    """The adaptive ladder (extension of bayes_kit/smc.py:42-43: next temperature = the largest step that keeps the ESS
    of the incremental weights at a fraction of the particles) on a conjugate Gaussian model with MANY observations --
    where the reference's t = n / N collapses at its first reweighting: every reweighting keeps its ESS, the ladder
    ends at exactly 1, and the particles carry the exact posterior (known in closed form)."""
    import torch

    rng = np.random.default_rng(seed)
    mu_true = rng.normal(size=D)
    ybar = torch.tensor(mu_true + rng.normal(size=D) / np.sqrt(n_obs), dtype=torch.float64)
    dev = ops.device
    ybar_d = ybar.to(dev)
    log_prior = lambda Th: -0.5 * (Th * Th).sum(dim=1)                           # N(0, I)
    log_lik = lambda Th: -0.5 * n_obs * ((Th - ybar_d[None, :]) ** 2).sum(dim=1)   # n_obs observations of N(theta, 1)
    model = bk.TorchPriorLikelihoodModel(log_prior, log_lik, D)
    init = rng.normal(size=(M, D))
    # (the reference's ladder with an affordable N: the first reweighting keeps a handful of particles)
    fixed = bk.TemperedLikelihoodSMC(model, M, 8, init, bk.metropolis_kernel(0.05), seed=seed, ops=ops)
    fixed.transition(1)
    assert fixed.last_ess < 0.05 * M
    smc = bk.TemperedLikelihoodSMC(model, M, 8, init, bk.hmc_kernel(0.7, 3, adapt_metric=True), seed=seed, ops=ops,
                                   adaptive=0.5)
    smc.run()
    T = np.array(smc.temperatures)
    assert T[-1] == 1.0 and smc.t == 1.0 and np.all(np.diff(T) > 0) and 5 < len(T) < 400
    ess = np.array(smc.ess_history)
    assert np.all(ess[:-1] >= 0.5 * M * (1 - 1e-3)) and ess[-1] >= 0.5 * M * (1 - 1e-3)  # (the last step may be shorter)
    assert np.all(ess[:-1] <= 0.5 * M * 1.05)   # ... and each step is as long as the target allows
    assert smc.time(0) == 0.0 and smc.time(1) == T[0] and smc.time(len(T)) == 1.0
    th = np.asarray(smc.thetas.cpu())
    post_var = 1.0 / (1.0 + n_obs)
    post_mean = ybar.numpy() * n_obs * post_var
    z = (th.mean(axis=0) - post_mean) / np.sqrt(post_var / (M / 4))
    assert np.abs(z).max() < 4.0, z
    np.testing.assert_allclose(th.var(axis=0, ddof=1), post_var, rtol=0.25)
    assert min(smc.kernel.accept_rates) > 0.3   # the adapted metric keeps the moves alive along the whole ladder
    return smc


def check_logistic_retemper(ops, N=3000, D=12, C=50):
    """bk.LogisticRegression.bk_retemper: (logp, grad) at another temperature from the untempered parts of an earlier
    evaluation equal a fresh evaluation at that temperature (same arithmetic in the finishing kernel: bit for bit)."""
    import torch

    g = torch.Generator().manual_seed(3)
    X = torch.randn((N, D), dtype=torch.float64, generator=g) / D ** 0.5
    y = (torch.rand(N, dtype=torch.float64, generator=g) < 0.4).to(torch.float64)
    m = bk.LogisticRegression(X.to(ops.device), y.to(ops.device), prior_scale=0.7, ops=ops)
    th = (torch.randn((D, C), dtype=torch.float64, generator=g) * 0.3).to(ops.device)
    f64 = dict(dtype=torch.float64, device=ops.device)
    g1, lp1, ll, gll = torch.empty_like(th), torch.empty(C, **f64), torch.empty(C, **f64), torch.empty_like(th)
    m.bk_eval(th, g1, lp1, 0.25, ll, gll)
    for t in (0.0, 0.25, 0.6, 1.0):
        gf, lf = torch.empty_like(th), torch.empty(C, **f64)
        m.bk_eval(th, gf, lf, t)
        gr, lr = torch.empty_like(th), torch.empty(C, **f64)
        m.bk_retemper(th, gll, ll, t, gr, lr)
        assert torch.equal(gr, gf), t
        torch.testing.assert_close(lr, lf, rtol=1e-13, atol=1e-13)  # (the fresh call sums 256 segment partials, this one has their sum)


class LegacyReplay:
    """The values a reference SMC run took from numpy's global stream (tests/golden/smc_*.npz), handed out again
    through RandomState's method names in the same order."""

    def __init__(self, z):
        self._z, self._u, self._c = z["normals"], z["uniforms"].reshape(-1), z["choice_uniforms"]
        self._D = self._z.shape[-1]
        self._z = self._z.reshape(-1, self._D)
        self._iz = self._iu = self._ic = 0

    def standard_normal(self, size):
        assert size == self._D
        self._iz += 1
        return self._z[self._iz - 1]

    def uniform(self):
        self._iu += 1
        return self._u[self._iu - 1]

    def random_sample(self, n):
        self._ic += 1
        assert n == self._c.shape[1]
        return self._c[self._ic - 1]


def check_smc_reference_stream(ops, name, source):
    """bayes_kit/smc.py:12-89 under np.random.seed(s), through the product: TemperedLikelihoodSMC in reference-stream
    mode against the fixture of the REAL reference run -- the moved particles, the ancestor indices and the resampled
    particles after every temperature, all bit-exact."""
    import torch

    from tests.helpers import load_case, smc_expected_thetas, smc_model

    case, z = load_case(name)
    M, N, spec = case["M"], case["N"], case["model"]
    if spec["kind"] == "ref_binomial":
        model = smc_model(spec)            # a reference-style host model: one particle per call (smc.py:28-32)
    else:
        om = smc_model(spec)               # the same densities as PyTorch functions of all particles at once
        dev = ops.device
        y, prec = (torch.as_tensor(v, dtype=torch.float64, device=dev) for v in (om._y, om._prec))
        c0 = om._c0

        def log_prior(Th):
            return c0 * (Th * Th).sum(dim=1)

        def log_lik(Th):
            r = Th - y
            return -0.5 * (prec * (r * r)).sum(dim=1)

        model = bk.TorchPriorLikelihoodModel(log_prior, log_lik, spec["D"])
    if source == "np.random":
        np.random.seed(case["seed"])
        stream = np.random
    elif source == "RandomState":
        stream = np.random.RandomState(case["seed"])
    else:
        stream = LegacyReplay(z)
    smc = bk.TemperedLikelihoodSMC(model, M, N, z["theta0"], bk.metropolis_kernel(case["scale"]), seed=stream, ops=ops)
    want = smc_expected_thetas(z)
    for n in range(1, N + 1):
        # the moved particles are what the resampling gathers from: read them between the two halves of transition()
        seen = {}
        gather = ops.gather_columns

        def spy(index, src, dst, _seen=seen, _gather=gather):
            _seen["moved"] = np.asarray(src.cpu()).T.copy()
            _gather(index, src, dst)

        ops.gather_columns = spy
        try:
            smc.transition(n)
        finally:
            del ops.gather_columns   # (the instance attribute: the class's method is visible again)
        assert np.array_equal(seen["moved"], z["moved"][n - 1]), (name, source, n)
        assert np.array_equal(np.asarray(smc._idx.cpu()), z["idx"][n - 1]), (name, source, n)
        assert np.array_equal(np.asarray(torch.as_tensor(smc.thetas).cpu()), want[n - 1]), (name, source, n)
    if source == "np.random":
        st = np.random.get_state(legacy=False)
        assert st["state"]["pos"] == int(z["final_pos"]) and st["has_gauss"] == int(z["final_has_gauss"])
        assert np.array_equal(st["state"]["key"][:8], z["final_key"])
    return smc


def check_funnel_vs_canonical_oracle(name, ops, model_factory=None, **extra):
    """The funnel fixtures' configurations, HIP against the ORACLE WITH THE LIBRARY'S SUMMATION ORDER AND THE LIBRARY'S
    exp (oracle.models.FunnelCanonical: 16-class sums of csrc/bk_lanes.hpp, bk_exp of include/bkhip_math.h restated in
    oracle/rng.py): the two run the same sequence of rounded operations, so theta and the momentum are required to be
    BIT-IDENTICAL over all draws (60 of funnel11_k3, 40 of funnel101_cfg4), the joint log density to 1e-12 (the
    kinetic-energy sum's order), the stream state exact.  SURVEY 8c asks for rel 1e-9; this is zero.  (Against the
    reference's own np.dot + libm exp the chaotic flow amplifies last-bit differences: that comparison, with its
    widening bound, is the CPU test of this oracle against the golden, tests/test_oracle_golden.py.)"""
    from oracle import models as om
    from tests.helpers import oracle_sampler, rng_state_words

    case, z = load_case(name)
    N, C, D = z["draws"].shape
    assert case["model"]["kind"] == "funnel"
    model = (model_factory or product_model)(case["model"], ops)
    s = build_sampler(case, model, ops, case["seed"], chains=C, **extra)
    oracles = [oracle_sampler(case, c, model=om.FunnelCanonical(D)) for c in range(C)]
    for n in range(N):
        th, lp = s.sample()
        th, lp = th.cpu().numpy(), lp.cpu().numpy()
        for c, o in enumerate(oracles):
            oth, olp = o.sample()
            assert np.array_equal(th[c], oth), (name, c, n, float(np.abs(th[c] - oth).max()))
            np.testing.assert_allclose(lp[c], olp, rtol=LOGP_RTOL, atol=1e-12, err_msg=f"{name} chain {c} draw {n} logp")
    st = s.rng_state().T
    rho = s._rho.cpu().numpy()
    for c, o in enumerate(oracles):
        np.testing.assert_array_equal(st[c], rng_state_words(o._rng))
        assert np.array_equal(rho[c], o._rho), (name, c)
    return s
