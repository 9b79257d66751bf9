"""HIP kernels vs the oracle / NumPy on a real MI355X, called through the C ABI."""
import math

import numpy as np
import pytest
import torch

import bayes_kit_amd as bk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bayes_kit_amd import _lib

    return _lib.default_ops()


def dev(a, ops):
    return torch.as_tensor(a).to(ops.device)


def make_state(seed, C, ops, chain0=0):
    from bayes_kit_amd._engine import make_streams

    return make_streams(seed, C, chain0, False, ops.device)


def test_native_library_is_the_loaded_one(ops):
    import os
    from bayes_kit_amd import _lib

    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.lib_path()) in maps


@pytest.mark.parametrize("wave_per_chain", [False, True])
def test_device_rng_matches_numpy_bit_for_bit(ops, wave_per_chain):
    from bayes_kit_amd import _lib

    C, D = 192, 3000  # 576k normals: ~140 tail draws, ~8500 wedge draws
    kind, st = make_state(991, C, ops, chain0=5)
    out = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    kin = torch.empty(C, dtype=torch.float64, device=ops.device)
    # wave_per_chain: k_zig_parallel (256 stream words of one chain per wavefront) + transpose
    work = ops.refresh_work(C, D) if wave_per_chain else None
    ops.momentum_refresh(kind, st, None, 0.0, 1.0, out, None, kin, None, work)
    logu = torch.empty(C, dtype=torch.float64, device=ops.device)
    ops.log_uniform(kind, st, logu)
    got, kin, logu = out.cpu().numpy(), kin.cpu().numpy(), logu.cpu().numpy()
    words = st.cpu().numpy().view(np.uint64)
    for c in range(C):
        g = np.random.Generator(np.random.Philox(key=[991, 5 + c]))
        ref = g.normal(size=D)
        assert np.array_equal(ref.view(np.uint64), got[:, c].view(np.uint64)), c
        np.testing.assert_allclose(kin[c], 0.5 * np.dot(ref, ref), rtol=1e-13)
        np.testing.assert_allclose(logu[c], np.log(g.uniform()), rtol=4e-16, atol=0)
        s = g.bit_generator.state
        assert [int(v) for v in s["state"]["counter"]] == [int(v) for v in words[2:6, c]]
        assert [int(v) for v in s["buffer"]] == [int(v) for v in words[6:10, c]]
        assert s["buffer_pos"] == int(words[10, c])


def test_device_rng_pcg64_and_loc_scale(ops):
    from bayes_kit_amd import _lib
    from bayes_kit_amd._engine import make_streams

    gens = [np.random.default_rng(100 + c) for c in range(8)]
    kind, st = make_streams([np.random.default_rng(100 + c) for c in range(8)], 8, 0, False, ops.device)
    assert kind == _lib.RNG_PCG64
    D = 257
    loc = np.random.default_rng(0).normal(size=(D, 8))
    out = torch.empty((D, 8), dtype=torch.float64, device=ops.device)
    ops.momentum_refresh(kind, st, dev(loc, ops), math.sqrt(1 - 0.3), math.sqrt(0.3), out, None, None)
    got = out.cpu().numpy()
    for c in range(8):
        ref = gens[c].normal(loc=loc[:, c] * np.sqrt(1 - 0.3), scale=np.sqrt(0.3), size=D)
        assert np.array_equal(ref, got[:, c])
    # active mask: inactive chains neither draw nor write
    act = torch.tensor([1, 0, 1, 0, 0, 1, 1, 0], dtype=torch.uint8, device=ops.device)
    before = st.clone()
    out2 = torch.full((D, 8), 7.0, dtype=torch.float64, device=ops.device)
    ops.momentum_refresh(kind, st, None, 0.0, 1.0, out2, None, None, act)
    o2 = out2.cpu().numpy()
    for c in range(8):
        if act[c]:
            assert np.array_equal(o2[:, c], gens[c].normal(size=D))
        else:
            assert (o2[:, c] == 7.0).all()
            assert torch.equal(st[:, c], before[:, c])


@pytest.mark.parametrize("C,D", [(512, 64), (130, 37), (1, 5), (77, 1), (4096, 128)])
@pytest.mark.parametrize("layout", ["chain", "dim", "odd"])
def test_kick_drift_bit_exact(ops, C, D, layout):
    rng = np.random.default_rng(C * 1000 + D)
    th, rho, g = (rng.normal(size=(D, C)) for _ in range(3))
    m = rng.uniform(0.5, 1.5, size=D)
    eps = 0.0123
    for (use_pre, pre, use_kick, kick, metric) in [(True, -0.5 * eps, True, eps, m), (False, 0.0, True, eps, m),
                                                   (True, 0.5 * eps, False, 0.0, None)]:
        t = (metric[:, None] * g) if metric is not None else g
        r = rho.copy()
        if use_pre:
            r = r + pre * t
        if use_kick:
            r = r + kick * t
        want_th = th + eps * r
        if layout == "chain":
            gd = dev(g, ops)
        elif layout == "dim":  # a row-major (C, D) model output viewed as [D, C]
            gd = dev(np.ascontiguousarray(g.T), ops).t()
            assert gd.stride() == (1, D) or D == 1 or C == 1
        else:  # arbitrary strides
            big = torch.zeros((D * 2 + 1, C * 3 + 1), dtype=torch.float64, device=ops.device)
            gd = big[1::2, 1::3][:D, :C]
            gd.copy_(dev(g, ops))
        th_d, rho_d = dev(th, ops), dev(rho, ops)
        tho, rhoo = torch.empty_like(th_d), torch.empty_like(rho_d)
        ops.kick_drift(th_d, tho, rho_d, rhoo, gd, None if metric is None else dev(metric, ops), eps, use_pre,
                       pre, use_kick, kick)
        assert np.array_equal(tho.cpu().numpy(), want_th)
        assert np.array_equal(rhoo.cpu().numpy(), r)
        # in place
        ops.kick_drift(th_d, th_d, rho_d, rho_d, gd, None if metric is None else dev(metric, ops), eps, use_pre,
                       pre, use_kick, kick)
        assert np.array_equal(th_d.cpu().numpy(), want_th) and np.array_equal(rho_d.cpu().numpy(), r)


def test_finish_gather_select_accept(ops):
    from bayes_kit_amd import _lib

    rng = np.random.default_rng(3)
    D, C = 45, 300
    rho, g, th = (rng.normal(size=(D, C)) for _ in range(3))
    m = rng.uniform(0.5, 1.5, size=D)
    half = 0.021
    for negate in (False, True):
        v = rho + half * (m[:, None] * g)
        if negate:
            v = -v
        kin = 0.5 * np.einsum("dc,dc->c", v, m[:, None] * v)
        out = torch.empty((D, C), dtype=torch.float64, device=ops.device)
        k = torch.empty(C, dtype=torch.float64, device=ops.device)
        ops.leapfrog_finish(dev(rho, ops), out, dev(g, ops), dev(m, ops), half, negate, k)
        assert np.array_equal(out.cpu().numpy(), v)
        np.testing.assert_allclose(k.cpu().numpy(), kin, rtol=1e-13)
    # gather first step
    idx = rng.permutation(C)[:97].astype(np.int32)
    tho = torch.empty((D, 128), dtype=torch.float64, device=ops.device)[:, :97]
    rhoo = torch.empty((D, 128), dtype=torch.float64, device=ops.device)[:, :97]
    ops.first_step_gather(dev(th, ops), dev(rho, ops), dev(g, ops), dev(idx, ops), tho, rhoo, dev(m, ops), 0.1, 0.05)
    r = rho[:, idx] + 0.05 * (m[:, None] * g[:, idx])
    assert np.array_equal(rhoo.cpu().numpy(), r)
    assert np.array_equal(tho.cpu().numpy(), th[:, idx] + 0.1 * r)
    # accept (both modes) + ballot popcount + select
    lp0, lp1, a0, a1 = (rng.normal(size=C) for _ in range(4))
    logu = np.log(rng.uniform(size=C))
    for mode in (_lib.ACCEPT_HMC, _lib.ACCEPT_MALA):
        if mode == _lib.ACCEPT_HMC:
            h0, h1 = lp0 - a0, lp1 - a1
            acc = logu < h1 - h0
            ret = np.where(acc, h1, h0)
        else:
            acc = logu < (lp1 - lp0) + (a1 - a0)
            ret = np.where(acc, lp1, lp0)
        lpc = dev(lp0.copy(), ops)
        mask = torch.empty(C, dtype=torch.uint8, device=ops.device)
        r_d = torch.empty(C, dtype=torch.float64, device=ops.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=ops.device)
        ops.mh_accept(mode, lpc, dev(a0, ops), dev(lp1, ops), dev(a1, ops), dev(logu, ops), mask, r_d, cnt)
        assert np.array_equal(mask.cpu().numpy().astype(bool), acc)
        assert np.array_equal(r_d.cpu().numpy(), ret)
        assert int(cnt.item()) == int(acc.sum())
        assert np.array_equal(lpc.cpu().numpy(), np.where(acc, lp1, lp0))
        dst0, dst1 = dev(th.copy(), ops), dev(rho.copy(), ops)
        ops.select_columns(mask, dst0, dev(g, ops), dst1, dev(th, ops))
        assert np.array_equal(dst0.cpu().numpy(), np.where(acc[None, :], g, th))
        assert np.array_equal(dst1.cpu().numpy(), np.where(acc[None, :], th, rho))


def test_targets_vs_oracle_models(ops):
    from oracle import models as om

    rng = np.random.default_rng(11)
    for D, C in [(1, 3), (16, 130), (101, 64)]:
        th = rng.normal(size=(D, C))
        lam = np.logspace(0, 2, D)
        cases = [("iso_gaussian", None, om.IsoGaussian(D), True), ("diag_gaussian", lam, om.DiagGaussian(lam), True)]
        if D > 1:
            cases.append(("funnel", None, om.Funnel(D), False))
        for kind, params, omodel, exact in cases:
            g = torch.empty((D, C), dtype=torch.float64, device=ops.device)
            lp = torch.empty(C, dtype=torch.float64, device=ops.device)
            p = None if params is None else dev(params, ops)
            ops.target_grad(kind, p, dev(th, ops), g, lp)
            g2 = torch.empty_like(g)
            ops.target_grad(kind, p, dev(th, ops), g2, None)  # gradient-only (streaming) form
            lp2 = torch.empty_like(lp)
            ops.target_grad(kind, p, dev(th, ops), None, lp2)
            assert torch.equal(g, g2) and torch.equal(lp, lp2)
            for c in range(C):
                olp, og = omodel.log_density_gradient(th[:, c])
                if exact:
                    assert np.array_equal(g[:, c].cpu().numpy(), og)
                    np.testing.assert_allclose(lp[c].item(), olp, rtol=1e-13)
                else:
                    np.testing.assert_allclose(g[:, c].cpu().numpy(), og, rtol=1e-13, atol=1e-300)
                    np.testing.assert_allclose(lp[c].item(), olp, rtol=1e-13)


def test_mala_kernels(ops):
    rng = np.random.default_rng(8)
    D, C = 33, 70
    th, g, gp = (rng.normal(size=(D, C)) for _ in range(3))
    eps = 0.07
    kind, st = make_state(55, C, ops)
    thp = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    ops.mala_propose(kind, st, dev(th, ops), dev(g, ops), thp, eps, math.sqrt(2 * eps))
    got = thp.cpu().numpy()
    for c in range(C):
        z = np.random.Generator(np.random.Philox(key=[55, c])).normal(size=D)
        assert np.array_equal(got[:, c], th[:, c] + eps * g[:, c] + np.sqrt(2 * eps) * z)
    f = torch.empty(C, dtype=torch.float64, device=ops.device)
    r = torch.empty(C, dtype=torch.float64, device=ops.device)
    ops.mala_logq(dev(th, ops), dev(g, ops), thp, dev(gp, ops), eps, f, r)
    xf = got - th - eps * g
    xr = th - got - eps * gp
    np.testing.assert_allclose(f.cpu().numpy(), (-0.25 / eps) * (xf * xf).sum(0), rtol=1e-13)
    np.testing.assert_allclose(r.cpu().numpy(), (-0.25 / eps) * (xr * xr).sum(0), rtol=1e-13)


def test_relayout(ops):
    rng = np.random.default_rng(1)
    for D, C in [(64, 64), (37, 130), (1, 9), (200, 3)]:
        a = rng.normal(size=(C, D))
        src = dev(a, ops).t()  # logical [D, C], dimension-contiguous
        dst = torch.empty((D, C), dtype=torch.float64, device=ops.device)
        ops.relayout(src, dst)
        assert np.array_equal(dst.cpu().numpy(), a.T)
        back = torch.empty((C, D), dtype=torch.float64, device=ops.device)
        ops.relayout(dst, back.t())
        assert np.array_equal(back.cpu().numpy(), a)


def test_compact_and_scatter(ops):
    rng = np.random.default_rng(4)
    for n in [1, 63, 64, 1000, 1024, 1025, 40000]:
        for p in [0.0, 0.03, 0.5, 1.0]:
            mask = (rng.uniform(size=n) < p).astype(np.uint8)
            idx = torch.full((n,), -1, dtype=torch.int32, device=ops.device)
            cnt = torch.zeros(1, dtype=torch.int32, device=ops.device)
            ops.compact_indices(dev(mask, ops), n, idx, cnt)
            want = np.nonzero(mask)[0]
            assert int(cnt.item()) == len(want)
            assert np.array_equal(idx.cpu().numpy()[: len(want)], want)
    D, C, n = 19, 500, 120
    src = [rng.normal(size=(D, 128)) for _ in range(3)]
    dst = [rng.normal(size=(D, C)) for _ in range(3)]
    ssrc, sdst = rng.normal(size=128), rng.normal(size=C)
    index = rng.permutation(C)[:n].astype(np.int32)
    mask = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    d_dst = [dev(a.copy(), ops) for a in dst]
    d_sdst = dev(sdst.copy(), ops)
    ops.scatter_columns(dev(mask, ops), dev(index, ops), n, d_dst, [dev(a, ops)[:, :n] for a in src], d_sdst,
                        dev(ssrc, ops))
    for a, b, d in zip(dst, src, d_dst):
        want = a.copy()
        want[:, index[mask.astype(bool)]] = b[:, :n][:, mask.astype(bool)]
        assert np.array_equal(d.cpu().numpy(), want)
    want = sdst.copy()
    want[index[mask.astype(bool)]] = ssrc[:n][mask.astype(bool)]
    assert np.array_equal(d_sdst.cpu().numpy(), want)


def test_dr_scalar_kernels_vs_fake(ops):
    """The bk_dr_* stage helpers against their NumPy specification (tests/fake_ops.py)."""
    from tests.fake_ops import FakeOps

    fake = FakeOps()
    rng = np.random.default_rng(12)
    C = 700
    logp, kin = rng.normal(size=C), rng.uniform(0, 5, size=C)

    def both(fn_name, tensors, *extra_before, **kw):
        pass

    f64 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64).copy())
    # begin
    outs = {}
    for name, o in (("hip", ops), ("fake", fake)):
        d = o.device
        H, h, rej = (torch.empty(C, dtype=torch.float64, device=d) for _ in range(3))
        alive = torch.empty(C, dtype=torch.uint8, device=d)
        o.dr_begin(f64(logp).to(d), f64(kin).to(d), H, h, rej, alive)
        outs[name] = [t.cpu().numpy() for t in (H, h, rej, alive)]
    for a, b in zip(outs["hip"], outs["fake"]):
        assert np.array_equal(a, b)
    # retry test / accept test share RNG streams: compare decisions and final stream positions
    from bayes_kit_amd._engine import make_streams

    rejv = np.log(rng.uniform(size=C))
    alive0 = (rng.uniform(size=C) < 0.7).astype(np.uint8)
    res = {}
    for name, o in (("hip", ops), ("fake", fake)):
        d = o.device
        kind, st = make_streams(31, C, 0, False, d)
        alive = torch.as_tensor(alive0.copy()).to(d)
        o.dr_retry_test(kind, st, f64(rejv).to(d), 1.0, alive)
        n = 300
        idx = torch.as_tensor(rng.permutation(C)[:n].astype(np.int32)) if name == "hip" else idx_keep
        idx_keep = idx
        a = f64(np.minimum(0, rng.normal(size=n))) if name == "hip" else a_keep
        a_keep = a
        Hn = f64(rng.normal(size=n)) if name == "hip" else Hn_keep
        Hn_keep = Hn
        cur_H, cur_h, rej2 = (f64(np.zeros(C)).to(d) for _ in range(3))
        acc = torch.empty(n, dtype=torch.uint8, device=d)
        o.dr_accept_test(kind, st, idx.to(d), a.to(d), Hn.to(d), n, cur_H, cur_h, rej2, alive, acc)
        res[name] = [alive.cpu().numpy(), acc.cpu().numpy(), cur_H.cpu().numpy(), cur_h.cpu().numpy(),
                     rej2.cpu().numpy(), st.cpu().numpy()]
    for i, (a_, b_) in enumerate(zip(res["hip"], res["fake"])):
        if i in (3, 4):
            np.testing.assert_allclose(a_, b_, rtol=1e-14, atol=0)  # log1p(-exp(a)): device vs libm
        else:
            assert np.array_equal(a_, b_), i
    # ghost update + accept prob
    m, n = 150, 400
    ga = np.where(rng.uniform(size=m) < 0.2, 0.0, -rng.uniform(0.01, 3, size=m))
    sub = rng.permutation(n)[:m].astype(np.int32)
    h0, Hv, cH, ch = rng.normal(size=n) - 1, rng.normal(size=n), rng.normal(size=C), -rng.uniform(size=C)
    cidx = rng.permutation(C)[:n].astype(np.int32)
    res = {}
    for name, o in (("hip", ops), ("fake", fake)):
        d = o.device
        h = f64(h0).to(d)
        live = torch.ones(n, dtype=torch.uint8, device=d)
        a = torch.full((n,), 7.0, dtype=torch.float64, device=d)
        o.dr_ghost_update(f64(ga).to(d), torch.as_tensor(sub).to(d), m, h, live, a)
        for pr in (1.0, 0.0):
            a2 = a.clone()
            o.dr_accept_prob(f64(Hv).to(d), f64(cH).to(d), h, f64(ch).to(d), torch.as_tensor(cidx).to(d), pr, live,
                             a2, n)
            res.setdefault(name, []).append(a2.cpu().numpy())
        res[name] += [h.cpu().numpy(), live.cpu().numpy()]
    for a_, b_ in zip(res["hip"], res["fake"]):
        np.testing.assert_allclose(a_, b_, rtol=1e-14, atol=0)


@pytest.mark.parametrize("D,C", [(16, 16), (128, 128), (130, 200), (64, 1000), (512, 4096), (37, 65), (3, 5)])
def test_dense_metric_apply_mfma(ops, D, C):
    """Y = M @ X on the fp64 matrix cores vs NumPy; asymmetric M and X catch any row/column or
    fragment-layout mix-up."""
    rng = np.random.default_rng(D * 7 + C)
    M = rng.normal(size=(D, D))
    X = rng.normal(size=(D, C))
    Y = torch.full((D, C), 7.0, dtype=torch.float64, device=ops.device)
    ops.dense_metric_apply(dev(M, ops), dev(X, ops), Y)
    want = M @ X
    np.testing.assert_allclose(Y.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * np.sqrt(D))
    # identity and a permutation matrix must reproduce X exactly
    P = np.eye(D)[rng.permutation(D)]
    ops.dense_metric_apply(dev(P, ops), dev(X, ops), Y)
    assert np.array_equal(Y.cpu().numpy(), P @ X)
    out = torch.empty(C, dtype=torch.float64, device=ops.device)
    ops.dot_columns(dev(X, ops), dev(want, ops), 0.5, out)
    np.testing.assert_allclose(out.cpu().numpy(), 0.5 * np.einsum("dc,dc->c", X, want), rtol=1e-13)


def test_resample_indices_and_gather(ops):
    """bk_resample_indices against numpy's own RandomState.choice(p = w / w.sum()) fed the same uniforms (smc.py:73):
    indices BIT-EXACT, the cdf bit-exact (np.sum's pairwise pieces, np.cumsum's sequential chain)."""
    rng = np.random.default_rng(9)
    for n in [1, 5, 7, 8, 9, 127, 128, 129, 1000, 1024, 1025, 4096, 4097, 8191, 8192, 8193, 8199, 20000, 50000, 70001]:
        w = np.exp(rng.normal(size=n) * 3.0) if n % 2 else rng.uniform(0.0, 2.0, size=n)
        w[rng.uniform(size=n) < 0.1] = 0.0
        if w.sum() == 0:
            w[0] = 1.0
        p = w / w.sum()
        rs = np.random.RandomState(n)
        st = rs.get_state()
        want = rs.choice(n, size=n + 3, replace=True, p=p)
        rs.set_state(st)
        u = rs.random_sample(n + 3)   # the uniforms choice() drew
        cdf = torch.empty(n, dtype=torch.float64, device=ops.device)
        idx = torch.empty(n + 3, dtype=torch.int32, device=ops.device)
        ops.resample_indices(dev(w, ops), dev(u, ops), cdf, idx)
        assert np.array_equal(cdf.cpu().numpy(), np.cumsum(p)), n
        got = idx.cpu().numpy()
        assert np.array_equal(got, want), n
        assert (w[got] > 0).all()  # zero-weight particles are never chosen
    D, M = 7, 300
    src = rng.normal(size=(D, M))
    idx = rng.integers(0, M, size=M).astype(np.int32)
    dst = torch.empty((D, M), dtype=torch.float64, device=ops.device)
    ops.gather_columns(dev(idx, ops), dev(src, ops), dst)
    assert np.array_equal(dst.cpu().numpy(), src[:, idx])
    kind, st = make_state(77, 100, ops)
    out = torch.empty(100, dtype=torch.float64, device=ops.device)
    ops.uniform(kind, st, out)
    for c in (0, 57, 99):
        assert out[c].item() == np.random.Generator(np.random.Philox(key=[77, c])).uniform()


@pytest.mark.parametrize("D", [32, 63, 64, 65, 255, 256, 257, 1000])
def test_wave_per_chain_refresh_resumes_anywhere(ops, D):
    """k_zig_parallel must continue a stream from any buffer position, over many consecutive
    calls (state written back each time), with loc/scale and the kinetic energy, exactly like
    the one-lane-per-chain kernel and like numpy."""
    C = 67
    kind, st_a = make_state(31337, C, ops)
    _, st_b = make_state(31337, C, ops)
    u = torch.empty(C, dtype=torch.float64, device=ops.device)
    work = ops.refresh_work(C, D)
    m = dev(np.linspace(0.5, 1.5, D), ops)
    loc = dev(np.random.default_rng(D).normal(size=(D, C)), ops)
    gens = {c: np.random.Generator(np.random.Philox(key=[31337, c])) for c in (0, 33, 66)}
    for rep in range(5):
        for _ in range(rep):  # shift the buffer position between calls
            ops.uniform(kind, st_a, u)
            ops.uniform(kind, st_b, u)
            for g in gens.values():
                g.uniform()
        a = torch.empty((D, C), dtype=torch.float64, device=ops.device)
        b = torch.empty((D, C), dtype=torch.float64, device=ops.device)
        ka = torch.empty(C, dtype=torch.float64, device=ops.device)
        kb = torch.empty(C, dtype=torch.float64, device=ops.device)
        ops.momentum_refresh(kind, st_a, loc, 0.7, 0.3, a, m, ka, None, work)
        ops.momentum_refresh(kind, st_b, loc, 0.7, 0.3, b, m, kb, None, None)
        assert torch.equal(a, b) and torch.equal(ka, kb) and torch.equal(st_a, st_b), (D, rep)
        for c, g in gens.items():
            want = g.normal(loc=loc[:, c].cpu().numpy() * 0.7, scale=0.3, size=D)
            assert np.array_equal(a[:, c].cpu().numpy(), want), (D, rep, c)


@pytest.mark.parametrize("C,D,with_metric", [(67, 101, True), (4096, 101, False), (1000, 33, True), (130, 129, False),
                                              (64, 400, True), (50, 8, True)])
def test_dr_refresh_begin_equals_refresh_then_begin(ops, C, D, with_metric):
    """bk_dr_refresh_begin (generator launch + ONE launch: transpose, partial refresh in place, kinetic energy,
    start of the draw, first retry uniform) against bk_momentum_refresh with one lane per chain followed by
    bk_dr_begin_retry: momenta, energies, H / h / rej / alive, counters, draw counter and stream positions.
    (D = 8: below the wavefront-per-chain threshold -- the entry point falls back to the two calls.)"""
    kind, st_a = make_state(4242, C, ops)
    _, st_b = make_state(4242, C, ops)
    f64 = dict(dtype=torch.float64, device=ops.device)
    g = torch.Generator(device=ops.device)
    g.manual_seed(C + D)
    rho_a = torch.randn((D, C), generator=g, **f64)
    rho_b = rho_a.clone()
    logp = torch.randn(C, generator=g, **f64)
    m = dev(np.linspace(0.5, 1.5, D), ops) if with_metric else None
    work = ops.refresh_work(C, D)

    def outs():
        return (torch.empty(C, **f64), torch.empty(C, **f64), torch.empty(C, **f64), torch.empty(C, **f64),
                torch.empty(C, dtype=torch.uint8, device=ops.device),
                torch.full((5,), 9, dtype=torch.int32, device=ops.device),
                torch.full((1,), 3, dtype=torch.int64, device=ops.device))

    for rep in range(3):
        ka, Ha, ha, ra, la, ca, na = outs()
        kb, Hb, hb, rb, lb, cb, nb = outs()
        ops.dr_refresh_begin(kind, st_a, rho_a, -0.9, 0.4, rho_a, m, ka, work, logp, Ha, ha, ra, la, 1.0, ca, na)
        ops.momentum_refresh(kind, st_b, rho_b, -0.9, 0.4, rho_b, m, kb, None, None)
        ops.dr_begin_retry(kind, st_b, logp, kb, Hb, hb, rb, lb, 1.0, cb, nb)
        for x, y in zip((rho_a, ka, Ha, ha, ra, la, ca, na, st_a), (rho_b, kb, Hb, hb, rb, lb, cb, nb, st_b)):
            assert torch.equal(x, y), (C, D, rep)
        assert int(na) == 4 and int(ca.abs().sum()) == 0 and int(la.sum()) == C


@pytest.mark.parametrize("C,D", [(1, 1), (64, 9), (129, 17), (130, 8), (1024, 33)])
def test_select_with_fused_output_copy(ops, C, D):
    """copy0 = array 0 after the select, for every chain; with and without the second pair; the
    16-byte path (even C) and the scalar one (odd C); masks mixed inside a lane pair."""
    rng = np.random.default_rng(C * 100 + D)
    th, prop, g, gp = (rng.normal(size=(D, C)) for _ in range(4))
    for acc in (rng.random(C) < 0.5, np.zeros(C, bool), np.ones(C, bool)):
        mask = torch.as_tensor(acc.astype(np.uint8)).to(ops.device)
        for two in (False, True):
            d0, d1 = dev(th.copy(), ops), dev(g.copy(), ops)
            out = torch.full((D, C), np.nan, dtype=torch.float64, device=ops.device)
            if two:
                ops.select_columns(mask, d0, dev(prop, ops), d1, dev(gp, ops), out)
            else:
                ops.select_columns(mask, d0, dev(prop, ops), None, None, out)
            want = np.where(acc[None, :], prop, th)
            assert np.array_equal(d0.cpu().numpy(), want)
            assert np.array_equal(out.cpu().numpy(), want)
            assert np.array_equal(d1.cpu().numpy(), np.where(acc[None, :], gp, g) if two else g)


@pytest.mark.parametrize("C,D", [(1, 1), (64, 9), (129, 17), (130, 8), (1024, 33)])
def test_blend_columns_writes_only_the_output(ops, C, D):
    """bk_blend_columns: out = mask ? b : a with both inputs left alone (16-byte and scalar paths, masks
    mixed inside a lane pair, all / none accepted)."""
    rng = np.random.default_rng(C * 7 + D)
    a, b = rng.normal(size=(D, C)), rng.normal(size=(D, C))
    for acc in (rng.random(C) < 0.5, np.zeros(C, bool), np.ones(C, bool)):
        mask = torch.as_tensor(acc.astype(np.uint8)).to(ops.device)
        da, db = dev(a, ops), dev(b, ops)
        out = torch.full((D, C), np.nan, dtype=torch.float64, device=ops.device)
        ops.blend_columns(mask, da, db, out)
        assert np.array_equal(out.cpu().numpy(), np.where(acc[None, :], b, a))
        assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(db.cpu().numpy(), b)


@pytest.mark.parametrize("C,D", [(5, 32), (70, 100), (64, 257), (9000, 40), (20000, 101), (33000, 33), (40000, 200)])
def test_chain_major_normals_and_transposing_proposal(ops, C, D):
    """bk_normals_chain_major leaves the normals chain-major (64, 32 or 16 lanes per chain depending on
    the launch: 32 lanes for the 9000- and 20,000-chain shapes, 16 for the last two); the MALA proposal kernel reading them
    through LDS tiles must equal the one reading the state layout, and numpy.  The optional snapshot
    is the table as it was before the call."""
    kind, st_a = make_state(4242, C, ops)
    _, st_b = make_state(4242, C, ops)
    dp = (D + 7) // 8 * 8
    zt = torch.full((C, dp), np.nan, dtype=torch.float64, device=ops.device)
    z = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    snap = torch.zeros_like(st_a)
    for _ in range(2):
        before = st_a.clone()
        ops.normals_chain_major(kind, st_a, zt, D, snap)
        ops.momentum_refresh(kind, st_b, None, 0.0, 1.0, z, None, None)
        assert torch.equal(zt[:, :D].t(), z) and torch.equal(st_a, st_b) and torch.equal(snap, before)
    g = np.random.Generator(np.random.Philox(key=[4242, C - 1]))
    g.normal(size=D)
    assert np.array_equal(zt[C - 1, :D].cpu().numpy(), g.normal(size=D))
    rng = np.random.default_rng(3)
    th, gr = rng.normal(size=(D, C)), rng.normal(size=(D, C))
    pa = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    pb = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    ops.mala_propose_from_normals(dev(th, ops), dev(gr, ops), zt[:, :D].t(), pa, 0.03, math.sqrt(0.06))
    ops.mala_propose_from_normals(dev(th, ops), dev(gr, ops), z, pb, 0.03, math.sqrt(0.06))
    assert torch.equal(pa, pb)
    assert np.array_equal(pa.cpu().numpy(), (th + 0.03 * gr) + math.sqrt(0.06) * z.cpu().numpy())


def test_differential_soak_against_the_cpu_stand_in(ops):
    """Random shapes, padded leading dimensions (views of wider buffers, as the DRGHMC lane sets and
    chain tiles use), column offsets that break 16-byte alignment, every gradient layout: each
    elementwise kernel must agree bit for bit with tests/fake_ops.py (NumPy) on the same inputs."""
    from tests.fake_ops import FakeOps

    fake = FakeOps()
    rng = np.random.default_rng(20260101)

    def pair(D, C, pad=None, off=None):
        """(device view, host view) of equal values and equal strides: [D, C] inside [D, off+C+pad]."""
        pad = int(rng.choice([0, 0, 1, 3, 64])) if pad is None else pad
        off = int(rng.choice([0, 0, 1, 2])) if off is None else off
        base = rng.normal(size=(D, off + C + pad))
        h = torch.from_numpy(base.copy())
        d = h.to(ops.device)
        return d[:, off:off + C], h[:, off:off + C], pad, off

    def same_layout(D, C, pad, off):
        return pair(D, C, pad, off)[:2]

    def eq(a, b, what):
        assert np.array_equal(a.cpu().numpy(), b.numpy()), what

    for it in range(400):
        C, D = int(rng.integers(1, 400)), int(rng.integers(1, 70))
        th_d, th_h, pad, off = pair(D, C)
        rho_d, rho_h = same_layout(D, C, pad, off)
        tho_d, tho_h = same_layout(D, C, pad, off)
        rhoo_d, rhoo_h = same_layout(D, C, pad, off)
        metric = rng.random(D) + 0.5 if rng.random() < 0.6 else None
        m_d = None if metric is None else torch.from_numpy(metric).to(ops.device)
        m_h = None if metric is None else torch.from_numpy(metric)
        # gradient in one of three layouts: the state layout, row-major (C, D) seen transposed, odd strides
        lay = it % 3
        if lay == 0:
            g_d, g_h = same_layout(D, C, pad, off)
        elif lay == 1:
            gh = torch.from_numpy(rng.normal(size=(C, D)))
            g_d, g_h = gh.to(ops.device).t(), gh.t()
        else:
            gh = torch.from_numpy(rng.normal(size=(D, 2 * C + 1)))
            g_d, g_h = gh.to(ops.device)[:, ::2][:, :C], gh[:, ::2][:, :C]
        eps, pre, kick = float(rng.normal()), float(rng.normal()), float(rng.normal())
        up, uk = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        ops.kick_drift(th_d, tho_d, rho_d, rhoo_d, g_d, m_d, eps, up, pre, uk, kick)
        fake.kick_drift(th_h, tho_h, rho_h, rhoo_h, g_h, m_h, eps, up, pre, uk, kick)
        eq(tho_d, tho_h, ("kick_drift theta", it, C, D, pad, off, lay))
        eq(rhoo_d, rhoo_h, ("kick_drift rho", it, C, D, pad, off, lay))

        # finish: half step (+ negate) and kinetic energy
        neg = bool(rng.integers(0, 2))
        kin_d = torch.empty(C, dtype=torch.float64, device=ops.device)
        kin_h = torch.empty(C, dtype=torch.float64)
        ops.leapfrog_finish(rho_d, rhoo_d, g_d, m_d, eps, neg, kin_d)
        fake.leapfrog_finish(rho_h, rhoo_h, g_h, m_h, eps, neg, kin_h)
        eq(rhoo_d, rhoo_h, ("finish rho", it, C, D, pad, off, lay))
        np.testing.assert_allclose(kin_d.cpu().numpy(), kin_h.numpy(), rtol=1e-13)

        # gather-first-step over a random lane set into a dense buffer, then scatter some back
        n = int(rng.integers(1, C + 1))
        idx = np.sort(rng.choice(C, size=n, replace=False)).astype(np.int32)
        idx_d, idx_h = torch.from_numpy(idx).to(ops.device), torch.from_numpy(idx)
        gth_d, gth_h, p2, o2 = pair(D, n)
        grh_d, grh_h = same_layout(D, n, p2, o2)
        gs_d, gs_h = same_layout(D, C, pad, off)  # gradient at the source, state layout
        ops.first_step_gather(th_d, rho_d, gs_d, idx_d, gth_d, grh_d, m_d, eps, pre)
        fake.first_step_gather(th_h, rho_h, gs_h, idx_h, gth_h, grh_h, m_h, eps, pre)
        eq(gth_d, gth_h, ("gather theta", it, C, D, n))
        eq(grh_d, grh_h, ("gather rho", it, C, D, n))
        acc = (rng.random(n) < 0.5).astype(np.uint8)
        acc_d, acc_h = torch.from_numpy(acc).to(ops.device), torch.from_numpy(acc)
        sd_d, sd_h = torch.zeros(C, dtype=torch.float64, device=ops.device), torch.zeros(C, dtype=torch.float64)
        ss = torch.from_numpy(rng.normal(size=n))
        ops.scatter_columns(acc_d, idx_d, n, [tho_d, rhoo_d], [gth_d, grh_d], sd_d, ss.to(ops.device))
        fake.scatter_columns(acc_h, idx_h, n, [tho_h, rhoo_h], [gth_h, grh_h], sd_h, ss)
        eq(tho_d, tho_h, ("scatter theta", it))
        eq(rhoo_d, rhoo_h, ("scatter rho", it))
        eq(sd_d, sd_h, ("scatter scalar", it))

        # masked select with the fused output copy
        mask = (rng.random(C) < rng.random()).astype(np.uint8)
        mk_d, mk_h = torch.from_numpy(mask).to(ops.device), torch.from_numpy(mask)
        out_d, out_h = same_layout(D, C, pad, off)
        two = bool(rng.integers(0, 2))
        ops.select_columns(mk_d, th_d, tho_d, rho_d if two else None, rhoo_d if two else None, out_d)
        fake.select_columns(mk_h, th_h, tho_h, rho_h if two else None, rhoo_h if two else None, out_h)
        eq(th_d, th_h, ("select dst0", it, C, D, pad, off))
        eq(rho_d, rho_h, ("select dst1", it))
        eq(out_d, out_h, ("select copy", it))

        # MALA proposal from normals in either layout; relayout; column gather
        zt = torch.from_numpy(rng.normal(size=(C, D + int(rng.integers(0, 5)))))
        z_d, z_h = (zt.to(ops.device)[:, :D].t(), zt[:, :D].t()) if it % 2 else same_layout(D, C, pad, off)
        ops.mala_propose_from_normals(th_d, gs_d, z_d, tho_d, 0.03, 0.2)
        fake.mala_propose_from_normals(th_h, gs_h, z_h, tho_h, 0.03, 0.2)
        eq(tho_d, tho_h, ("mala propose", it, C, D, pad, off))
        ops.relayout(g_d, rhoo_d)
        fake.relayout(g_h, rhoo_h)
        eq(rhoo_d, rhoo_h, ("relayout", it, lay))
        dst_d = torch.empty((D, n), dtype=torch.float64, device=ops.device)
        dst_h = torch.empty((D, n), dtype=torch.float64)
        ops.gather_columns(idx_d, th_d, dst_d)
        fake.gather_columns(idx_h, th_h, dst_h)
        eq(dst_d, dst_h, ("gather_columns", it))


def test_accept_with_non_finite_log_densities(ops):
    """IEEE semantics of `log(u) < ratio` (hmc.py:60, metropolis.py:70-76): NaN ratios reject,
    +inf accepts, -inf rejects, inf - inf = NaN rejects -- as NumPy evaluates the same expressions."""
    from bayes_kit_amd import _lib

    inf, nan = np.inf, np.nan
    lp0 = np.array([0.0, 0.0, 0.0, -inf, inf, nan, 0.0, -inf, 1.0])
    lp1 = np.array([nan, inf, -inf, -inf, inf, 0.0, 0.5, 0.0, 1.0])
    a0 = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, nan, 0.0, inf])
    a1 = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, inf])
    logu = np.full(9, -0.3)
    for mode in (_lib.ACCEPT_HMC, _lib.ACCEPT_MALA):
        with np.errstate(invalid="ignore"):
            if mode == _lib.ACCEPT_HMC:
                want = logu < (lp1 - a1) - (lp0 - a0)
            else:
                want = logu < (lp1 - lp0) + (a1 - a0)
        mask = torch.empty(9, dtype=torch.uint8, device=ops.device)
        cur = dev(lp0.copy(), ops)
        ops.mh_accept(mode, cur, dev(a0, ops), dev(lp1, ops), dev(a1, ops), dev(logu, ops), mask, None, None)
        assert mask.cpu().numpy().astype(bool).tolist() == want.tolist(), mode


@pytest.mark.parametrize("N,C", [(4, 3), (5, 70), (63, 17), (64, 33), (65, 16), (130, 100), (1000, 37), (1271, 20),
                                 (1300, 9), (3000, 5), (6001, 3), (12000, 2),
                                 (287, 6), (288, 7), (289, 17), (575, 3), (576, 5), (577, 9), (1151, 4), (1153, 2),
                                 (5000, 3), (16383, 1)])
def test_lds_staged_ess_and_autocorr_vs_oracle(ops, N, C):
    """bk_ess / bk_autocorr with the series staged in LDS (16 / 8 / 4 / 2 / 1 chains per workgroup by N,
    one wavefront per chain; below 288 draws one lane per lag, from 288 on 64 lags per block from register tiles:
    chunks of 576 draws, so 287 .. 289, 575 .. 577, 1151 / 1153 sit on the seams) against oracle/diagnostics.py
    (the reference's FFT formulation): AR(1) chains of several persistences up to 0.999 (many 64-lag blocks),
    antithetic chains (negative first pair), both estimators, ragged last workgroup; autocorr() itself takes the
    FFT from 256 draws on, so the direct all-lag kernel is called by name as well."""
    from oracle import diagnostics as od

    rng = np.random.default_rng(N * 1000 + C)
    x = np.empty((N, C))
    for c in range(C):
        phi = [-0.6, 0.0, 0.5, 0.9, 0.98, 0.999][c % 6]
        e = rng.normal(size=N)
        x[0, c] = e[0]
        for t in range(1, N):
            x[t, c] = phi * x[t - 1, c] + e[t]
    xd = torch.from_numpy(x).to(ops.device)
    ac = bk.autocorr(xd).cpu().numpy()
    direct = torch.empty_like(xd)
    ops.autocorr(xd, direct)
    direct = direct.cpu().numpy()
    for c in range(C):
        want = od.autocorr(x[:, c])
        np.testing.assert_allclose(ac[:, c], want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(direct[:, c], want, rtol=0, atol=1e-12)
    for fn, ofn in ((bk.ess, od.ess), (bk.ess_ipse, od.ess_ipse), (bk.iat, od.iat), (bk.iat_ipse, od.iat_ipse)):
        got = fn(xd).cpu().numpy()
        want = np.array([ofn(x[:, c]) for c in range(C)])
        np.testing.assert_allclose(got, want, rtol=1e-9)


def test_long_chains_take_the_fft_formulation(ops):
    """N >= FFT_MIN_DRAWS: autocorrelation by the reference's own FFT formula (bk_autocorr_fft) + the library's
    Geyer scan kernel; against the oracle, and against the direct-sum kernel at the switch-over."""
    from oracle import diagnostics as od
    from bayes_kit_amd import diagnostics as dg

    rng = np.random.default_rng(3)
    N, C = 20000, 6
    x = np.empty((N, C))
    for c in range(C):
        phi = [0.3, 0.9, 0.99][c % 3]
        e = rng.normal(size=N)
        x[0, c] = e[0]
        for t in range(1, N):
            x[t, c] = phi * x[t - 1, c] + e[t]
    xd = torch.from_numpy(x).to(ops.device)
    assert N >= dg.FFT_MIN_DRAWS
    np.testing.assert_allclose(bk.ess(xd).cpu().numpy(), [od.ess(x[:, c]) for c in range(C)], rtol=1e-9)
    np.testing.assert_allclose(bk.autocorr(xd)[:, 1].cpu().numpy(), od.autocorr(x[:, 1]), rtol=0, atol=1e-12)
    # the two formulations agree where they meet
    short = xd[:12000].contiguous()
    direct = bk.ess(short).cpu().numpy()
    e2 = torch.empty(C, dtype=torch.float64, device=ops.device)
    ops.iat_from_acor(dg._autocorr_fft(short, ops), 0, e2, None)
    np.testing.assert_allclose(direct, e2.cpu().numpy(), rtol=1e-9)


@pytest.mark.parametrize("N,C", [(2, 1), (3, 2), (4, 5), (5, 64), (17, 3), (64, 129), (100, 7), (257, 66), (1000, 130),
                                 (4096, 9), (4097, 2), (16384, 5), (40000, 3)])
def test_autocorr_fft_against_numpy(ops, N, C):
    """bk_autocorr_fft (Stockham radix-8/4/2 passes across the rows, two real series per complex column, unit-variance
    scaling on the way in) against autocorr.py:23-33 evaluated with numpy.fft, for every transform size from 4 to
    131,072, odd and even column counts, columns of very different scale side by side, a strided input view."""
    rng = np.random.default_rng(N * 1000 + C)
    x = np.empty((N, C))
    for c in range(C):
        phi = [0.0, 0.5, 0.95, 0.999][c % 4]
        e = rng.normal(size=N)
        x[0, c] = e[0]
        for t in range(1, N):
            x[t, c] = phi * x[t - 1, c] + e[t]
        x[:, c] = (x[:, c] + 10.0 * (c % 3)) * 10.0 ** (3 * (c % 5) - 6)   # scales 1e-6 .. 1e6, offsets
    wide = torch.zeros((N, C + 3), dtype=torch.float64, device=ops.device)
    wide[:, :C] = torch.from_numpy(x).to(ops.device)
    out = torch.full((N, C + 1), float("nan"), dtype=torch.float64, device=ops.device)
    ops.autocorr_fft(wide[:, :C], out[:, :C])
    got = out[:, :C].cpu().numpy()
    assert torch.isnan(out[:, C]).all()
    size = 1 << int(np.ceil(np.log2(2 * N - 1)))
    for c in range(C):
        nd = x[:, c] - x[:, c].mean()
        want = np.fft.ifft(np.abs(np.fft.fft(nd, size)) ** 2).real[:N] / np.var(x[:, c]) / N
        np.testing.assert_allclose(got[:, c], want, rtol=0, atol=2e-12, err_msg=str((N, C, c)))
    assert np.allclose(got[0], 1.0, rtol=0, atol=1e-13)


def test_autocorr_fft_constant_series_gives_nan_like_the_reference(ops):
    """np.var == 0: the reference divides 0 by 0 (autocorr.py:32); so does the kernel, for that column only."""
    x = torch.randn((50, 4), dtype=torch.float64, device=ops.device)
    x[:, 2] = 3.0
    out = torch.empty_like(x)
    ops.autocorr_fft(x, out)
    assert torch.isnan(out[:, 2]).all() and torch.isfinite(out[:, [0, 1, 3]]).all()


def test_sort_entry_points_and_pooled_ranks(ops):
    """bk_sort_by_key (stable, -0.0 < +0.0, duplicates, infinities), bk_count_below, bk_scatter_ranks and
    the pooled ranks built on them, against torch's stable sort / NumPy."""
    rng = np.random.default_rng(17)
    for n in (1, 2, 63, 1000, 4095, 4096, 4097, 70_000, 3_000_001):
        v = rng.normal(size=n)
        if n == 70_000:
            v = rng.integers(0, 8, size=n).astype(np.float64)   # a few distinct keys: most passes see ONE digit
        elif n >= 63:
            v[::7] = np.round(v[::7])          # many exact ties
            v[3], v[5], v[11], v[12] = 0.0, -0.0, np.inf, -np.inf
        keys = torch.from_numpy(v).to(ops.device)
        pay = torch.arange(n, dtype=torch.int64, device=ops.device)
        ks, ps = ops.sort_by_key(keys, pay)
        ref = torch.sort(keys, stable=True)
        assert torch.equal(torch.sort(ps).values, pay) and torch.equal(keys[ps], ks)   # a permutation, pairs intact
        assert torch.equal(ks, ref.values) or n < 63
        assert torch.equal(ks.abs(), ref.values.abs()) and bool((ks[1:] >= ks[:-1]).all())
        if n < 63:
            assert torch.equal(ps, ref.indices)
        else:  # stable within equal keys (radix order puts -0.0 before +0.0, torch treats them as equal)
            same = ks[1:] == ks[:-1]
            signed_same = same & (torch.signbit(ks[1:]) == torch.signbit(ks[:-1]))
            assert bool((ps[1:][signed_same] > ps[:-1][signed_same]).all())
        q = torch.from_numpy(np.array([-np.inf, -1.0, 0.0, 0.5, np.inf])).to(ops.device)
        np.testing.assert_array_equal(ops.count_below(ks, q).cpu().numpy(), np.searchsorted(np.sort(v), q.cpu().numpy(), side="left"))
        ranks = torch.empty(n, dtype=torch.float64, device=ops.device)
        ops.scatter_ranks(ps, 10.0, ranks)
        assert torch.equal(ranks[ps], torch.arange(11, n + 11, dtype=torch.float64, device=ops.device))
    x = torch.from_numpy(rng.normal(size=(200, 37))).to(ops.device)
    from bayes_kit_amd import diagnostics as dg

    r = dg.rank_chains(x).cpu().numpy()
    flat = x.t().contiguous().reshape(-1).cpu().numpy()
    want = (flat.argsort(kind="stable").argsort() + 1).reshape(37, 200).T
    np.testing.assert_array_equal(r, want)


def test_sample_sort_and_rhat_collectives_run_on_rccl(ops):
    """The cross-rank code paths on the real collective library: a one-rank `nccl` (= RCCL) process group
    on the GPU -- all_gather, all_to_all_single and the gathered partial sums run through RCCL with
    device tensors; results equal the no-group paths."""
    import socket

    import torch.distributed as dist
    from bayes_kit_amd import diagnostics as dg

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=ops.device)
    try:
        assert dist.get_backend() == "nccl"
        rng = np.random.default_rng(2)
        x = torch.from_numpy(rng.normal(size=(300, 64))).to(ops.device)
        x[::5] = torch.round(x[::5])
        got = dg._ranks_pooled_across_ranks(x, ops)
        assert torch.equal(got, dg.rank_chains(x))
        t = torch.ones(5, dtype=torch.float64, device=ops.device)
        dist.all_reduce(t)
        assert float(t.sum().item()) == 5.0
        np.testing.assert_allclose(bk.rhat(x), dg._rhat_of_columns(x, None, ops, None), rtol=0)
        mom = bk.RunningMoments(3, 64)
        for i in range(4):
            mom.update(x[i * 3:(i + 1) * 3].contiguous())
        assert np.isfinite(mom.rhat()).all()
        assert bk.dist.sum_over_ranks(2.5, ops.device) == 2.5
    finally:
        dist.destroy_process_group()


def test_device_bk_exp_equals_the_oracle_restatement_bit_for_bit():
    """bk_exp (include/bkhip_math.h) as the DEVICE evaluates it against oracle.rng.exp_bk (exact rational fma): read off the
    built-in funnel's gradient -- at theta = (v, -1) the entry of the row is -(bk_exp(-v) * -1) = bk_exp(-v) exactly."""
    from oracle.rng import exp_bk

    ops = bk._lib.default_ops()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.normal(size=6000) * 4.0, rng.uniform(-0.4, 0.4, 1500), rng.uniform(-746.5, 710.5, 3000),
                         rng.normal(size=500) * 1e-9,
                         [0.0, -0.0, 1e-300, -1e-300, 709.78, 709.782712893384, 709.7827128933841, 709.79, 710.0, 711.0, -745.13,
                          -745.1332191019412, -745.14, -746.0, -747.0, -708.4, -740.0, 1e308, -1e308, float("inf"), float("-inf"),
                          float("nan")]])
    th = torch.from_numpy(np.stack([-xs, -np.ones_like(xs)])).to(ops.device)
    grad, lp = torch.empty_like(th), torch.empty(th.shape[1], dtype=torch.float64, device=ops.device)
    ops.target_grad("funnel", None, th, grad, lp)
    got = grad[1].cpu().numpy()
    want = np.array([exp_bk(float(x)) for x in xs])
    assert np.array_equal(got.view(np.uint64)[:-1], want.view(np.uint64)[:-1])
    assert np.isnan(got[-1]) and np.isnan(want[-1])


def test_funnel_proposal_geometries_give_the_same_values():
    """bk_dr_proposal_funnel picks 4, 8 or 16 lanes of a wavefront per chain from the size of the lane set (on the
    host when it knows the size, on the device otherwise); the sum over a chain's coordinates has one canonical
    order, so the same chains integrated in sets of different sizes give the same bits -- and the same as the
    gradient op + kick/drift launches (step-by-step)."""
    ops = bk._lib.default_ops()
    dev = ops.device
    f64 = dict(dtype=torch.float64, device=dev)
    for D, steps, h in ((101, 7, 0.05), (18, 5, 0.1), (129, 3, 0.02)):
        C = 13000
        g = torch.Generator(device=dev)
        g.manual_seed(D)
        th = torch.randn((D, C), generator=g, **f64)
        th[0] *= 2.0
        rho = torch.randn((D, C), generator=g, **f64)
        grad, lp = torch.empty_like(th), torch.empty(C, **f64)
        ops.target_grad("funnel", None, th, grad, lp)
        metric = torch.linspace(0.8, 1.2, D, **f64) if D == 18 else None

        def run(n, n_on_device):
            idx = None if n == C else torch.arange(n, dtype=torch.int32, device=dev)
            out = [torch.full((D, C), float("nan"), **f64) for _ in range(3)]
            lpo, kin = torch.empty(C, **f64), torch.empty(C, **f64)
            n_dev = torch.tensor([n], dtype=torch.int32, device=dev) if n_on_device else None
            views = [o if n_on_device else o[:, :n] for o in out]
            ops.dr_proposal_funnel(th, rho, grad, idx, views[0], views[1], views[2], lpo if n_on_device else lpo[:n],
                                   kin if n_on_device else kin[:n], metric, h, steps, n_dev=n_dev)
            return [o[:, :n].clone() for o in out] + [lpo[:n].clone(), kin[:n].clone()]

        ref = run(C, False)  # 13,000 lanes known on the host: 4 lanes per chain
        for n, on_dev in ((C, True), (6000, False), (6000, True), (1000, False), (1000, True), (4608, True), (12288, True)):
            got = run(n, on_dev)
            for a, b in zip(got, ref):
                assert torch.equal(a, b[..., :n]), (D, n, on_dev)
        # ... and the step-by-step launches (first_step_gather, gradient op, kick + drift, finish)
        tho, rhoo, go = (torch.empty((D, 1000), **f64) for _ in range(3))
        lpo, kin = torch.empty(1000, **f64), torch.empty(1000, **f64)
        ops.first_step_gather(th, rho, grad, None, tho, rhoo, metric, h, 0.5 * h)
        for _ in range(steps - 1):
            ops.target_grad("funnel", None, tho, go, None)
            ops.kick_drift(tho, tho, rhoo, rhoo, go, metric, h, False, 0.0, True, h)
        ops.target_grad("funnel", None, tho, go, lpo)
        ops.leapfrog_finish(rhoo, rhoo, go, metric, 0.5 * h, True, kin)
        assert torch.equal(tho, ref[0][:, :1000]) and torch.equal(rhoo, ref[1][:, :1000]) and torch.equal(go, ref[2][:, :1000])
        assert torch.equal(lpo, ref[3][:1000])
        np.testing.assert_allclose(kin.cpu().numpy(), ref[4][:1000].cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("D,n,prob_retry", [(101, 13000, 1.0), (101, 5000, 0.0), (18, 900, 1.0), (129, 13000, 1.0), (2, 300, 1.0)])
def test_funnel_proposal_with_its_first_ghost_equals_two_launches(D, n, prob_retry):
    """bk_ghost0: a proposal launch that also integrates the first ghost of every lane it produces (from its
    registers) and applies it to the produced level, against the ghost as a launch of its own with a bk_ghost_link
    -- same level H / h / live / a, same list of lanes that go on (as a set), same lane statistics; for every
    lanes-per-chain geometry (13,000 / 5,000 / fewer lanes), sizes known on the host or on the device."""
    ops = bk._lib.default_ops()
    dev = ops.device
    f64 = dict(dtype=torch.float64, device=dev)
    C = 13000
    g = torch.Generator(device=dev)
    g.manual_seed(D + n)
    th = torch.randn((D, C), generator=g, **f64)
    th[0] *= 2.0
    rho = torch.randn((D, C), generator=g, **f64)
    grad, lp = torch.empty_like(th), torch.empty(C, **f64)
    ops.target_grad("funnel", None, th, grad, lp)
    metric = torch.linspace(0.8, 1.2, D, **f64) if D == 18 else None
    idx = torch.randperm(C, generator=torch.Generator().manual_seed(3))[:n].to(torch.int32).to(dev)
    h, steps, gh, gsteps = 0.3, 4, 0.9, 3   # a long first-kind step: many ghosts are accepted outright (g == 0)

    def level():
        out = [torch.full((D, n), float("nan"), **f64) for _ in range(3)]
        return out, torch.empty(n, **f64), torch.empty(n, **f64), (torch.empty(n, **f64), torch.empty(n, **f64),
                                                                  torch.empty(n, dtype=torch.uint8, device=dev))

    for on_dev in (False, True):
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev) if on_dev else None
        # two launches: the proposal, then its ghost with a link to it
        oa, lpa, kina, lva = level()
        a_par = torch.full((n,), 7.0, **f64)
        lanes_a = torch.zeros(2, dtype=torch.int32, device=dev)
        tot_a = torch.full((2,), 5, dtype=torch.int64, device=dev)
        ops.dr_proposal_funnel(th, rho, grad, idx, *oa, lpa, kina, metric, h, steps, n_dev=n_dev, lanes_out=lanes_a[0:1],
                               lanes_total=tot_a[0:1], level=lva)
        og, lpg, king, lvg = level()
        a_g = torch.empty(n, **f64)
        list_a, cnt_a = torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        link = ops.ghost_link(lva[0], lva[1], lva[2], a_par, a_g, prob_retry, list_a, cnt_a)
        ops.dr_proposal_funnel(oa[0], oa[1], oa[2], None, *og, lpg, king, metric, gh, gsteps, n_dev=n_dev,
                               lanes_out=lanes_a[1:2], lanes_total=tot_a[1:2], level=lvg, ghost=link)
        # one launch
        ob, lpb, kinb, lvb = level()
        b_par = torch.full((n,), 7.0, **f64)
        lanes_b = torch.zeros(2, dtype=torch.int32, device=dev)
        tot_b = torch.full((2,), 5, dtype=torch.int64, device=dev)
        list_b, cnt_b = torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        g0 = ops.ghost0(gh, gsteps, b_par, prob_retry, list_b, cnt_b, lanes_b[1:2], tot_b[1:2])
        ops.dr_proposal_funnel(th, rho, grad, idx, *ob, lpb, kinb, metric, h, steps, n_dev=n_dev, lanes_out=lanes_b[0:1],
                               lanes_total=tot_b[0:1], level=lvb, ghost0=g0)
        for x, y in zip(oa + [lpa, kina] + list(lva) + [a_par, lanes_a, tot_a, cnt_a],
                        ob + [lpb, kinb] + list(lvb) + [b_par, lanes_b, tot_b, cnt_b]):
            assert torch.equal(x, y), (D, n, on_dev)
        m = int(cnt_a)
        assert torch.equal(list_a[:m].sort().values, list_b[:m].sort().values)
        live = lva[2].bool()
        assert 0 < m == int(live.sum()) < n and torch.isinf(a_par[~live]).all() and (lva[1][live] <= 0).all() and (lva[1][live] < 0).any()


@pytest.mark.parametrize("D,n", [(101, 5000), (18, 13000), (40, 600)])
def test_funnel_ghost_with_its_own_ghost_applies_itself_to_its_parent(D, n):
    """A ghost level with ONE ghost of its own: proposal + that ghost (bk_ghost0) + the level's acceptance probability
    against its parent lanes and the parent's update (bk_ghost_link) in one launch, against the link as
    bk_dr_accept_prob_ghost_next in a launch of its own."""
    ops = bk._lib.default_ops()
    dev = ops.device
    f64 = dict(dtype=torch.float64, device=dev)
    C = 13000
    g = torch.Generator(device=dev)
    g.manual_seed(D * 11 + n)
    th = torch.randn((D, C), generator=g, **f64)
    th[0] *= 2.0
    rho = torch.randn((D, C), generator=g, **f64)
    grad, lp = torch.empty_like(th), torch.empty(C, **f64)
    ops.target_grad("funnel", None, th, grad, lp)
    metric = torch.linspace(0.8, 1.2, D, **f64) if D == 18 else None
    sub = torch.randperm(C, generator=torch.Generator().manual_seed(8))[:n].to(torch.int32).to(dev)
    kin0 = torch.empty(C, **f64)
    ops.leapfrog_finish(rho, None, None, metric, 0.0, False, kin0)
    h, steps, gh, gsteps, pr = 0.25, 4, 0.7, 3, 1.0

    def run(fused):
        par_H = (lp - kin0).clone() + 0.5
        par_h = -torch.rand(C, generator=torch.Generator().manual_seed(2), dtype=torch.float64).to(dev)
        par_live = torch.ones(C, dtype=torch.uint8, device=dev)
        par_a = torch.full((C,), 3.0, **f64)
        out = [torch.full((D, n), float("nan"), **f64) for _ in range(3)]
        lpo, kino = torch.empty(n, **f64), torch.empty(n, **f64)
        lv = (torch.empty(n, **f64), torch.empty(n, **f64), torch.empty(n, dtype=torch.uint8, device=dev))
        a = torch.full((n,), 5.0, **f64)
        nlist = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ncnt = torch.zeros(1, dtype=torch.int32, device=dev)
        g0 = ops.ghost0(gh, gsteps, a, pr)
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        kw = dict(n_dev=n_dev, level=lv, ghost0=g0)
        if fused:
            link = ops.ghost_link(par_H, par_h, par_live, par_a, a, pr, nlist, ncnt)
            ops.dr_proposal_funnel(th, rho, grad, sub, *out, lpo, kino, metric, h, steps, ghost=link, **kw)
        else:
            ops.dr_proposal_funnel(th, rho, grad, sub, *out, lpo, kino, metric, h, steps, **kw)
            ops.dr_accept_prob_ghost_next(lv[0], par_H, lv[1], par_h, sub, pr, lv[2], a, n, par_live, par_a, nlist, ncnt,
                                          n_dev=n_dev)
        m = int(ncnt)
        return out + [lpo, kino, *lv, a, par_H, par_h, par_live, par_a, ncnt, nlist[:m].sort().values]

    one, two = run(True), run(False)
    for i, (x, y) in enumerate(zip(one, two)):
        assert torch.equal(x, y), (D, n, i)
    assert 0 < int(one[-2]) < n and 0 < int(one[7].sum()) < n


def test_background_generator_launch_gives_the_same_stream():
    """bk_normals_chain_major: a bounded number of workgroups, each walking several groups of chains
    (a background kernel beside a streaming one), against the one-workgroup-per-group launch."""
    ops = bk._lib.default_ops()
    for C, D in ((1000, 70), (4096, 128), (333, 1000)):
        st_a = torch.zeros((bk._lib.RNG_WORDS, C), dtype=torch.int64, device=ops.device)
        ops.rng_init_philox(st_a, 99, 5)
        st_b = st_a.clone()
        dp = (D + 7) // 8 * 8
        za = torch.full((C, dp), float("nan"), dtype=torch.float64, device=ops.device)
        zb = za.clone()
        sa, sb = torch.empty_like(st_a), torch.empty_like(st_a)
        for rep in range(3):  # (resumes mid-buffer on the later calls)
            ops.normals_chain_major(bk._lib.RNG_PHILOX, st_a, za, D, sa)
            ops.normals_chain_major(bk._lib.RNG_PHILOX, st_b, zb, D, sb, max_workgroups=7)
            assert torch.equal(za[:, :D], zb[:, :D]) and torch.equal(st_a, st_b) and torch.equal(sa, sb), (C, D, rep)
