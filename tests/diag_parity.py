"""Diagnostics parity bodies shared by the CPU (fake ops) and GPU (HIP) tests."""
import os

import numpy as np
import pytest
import torch

import bayes_kit_amd as bk
from tests.helpers import GOLDEN


def check_special_values_in_ranks(ops):
    """rhat.py:51-52 ranks with numpy's comparison order: a NaN (whatever its sign bit) ranks LAST, -0.0 and
    +0.0 are one value.  The device sorts raw bit patterns (radix), so the keys are canonicalised first."""
    from bayes_kit_amd.rhat import rank_chains
    from oracle import diagnostics as od

    neg_nan = np.frombuffer(np.uint64(0xFFF8000000000001).tobytes(), dtype=np.float64)[0]
    assert np.isnan(neg_nan) and np.signbit(neg_nan)
    # (a) no ties: the oracle (numpy's default argsort) is the reference's behaviour
    a = np.array([[3.5, -0.0, 1.0, -2.0], [neg_nan, 7.0, -np.inf, 0.5], [np.inf, -1.0, 2.0, 9.0]])
    want = od.rank_chains([r for r in a])
    got = rank_chains([r for r in a], ops=ops)
    assert [list(r) for r in got] == [list(r) for r in want]
    assert got[1][0] == 12.0  # the NaN is last
    x = torch.from_numpy(np.ascontiguousarray(a.T)).to(ops.device)
    np.testing.assert_allclose(bk.rank_normalized_rhat(x, ops=ops), od.rank_normalized_rhat([r for r in a]), rtol=1e-12)
    # (b) ties: -0.0 / +0.0 rank in pooled order (numpy leaves tie order to its sort; the stable one is taken)
    b = np.array([[0.0, -0.0, 5.0], [-0.0, 0.0, -1.0], [1.0, -0.0, 0.0]])
    pooled = b.reshape(-1)
    stable = (np.argsort(np.argsort(pooled, kind="stable"), kind="stable") + 1).astype(np.float64).reshape(b.shape)
    got = rank_chains([r for r in b], ops=ops)
    assert [list(r) for r in got] == [list(r) for r in stable]
    assert sorted(np.asarray(got).reshape(-1)[pooled == 0.0]) == [2.0, 3.0, 4.0, 5.0, 6.0, 7.0]


def check_diagnostics(ops, ess_rtol):
    check_special_values_in_ranks(ops)
    z = np.load(os.path.join(GOLDEN, "diagnostics.npz"))
    chains = list(z["rhat_chains"])
    # reference-style inputs (list of 1-D chains)
    np.testing.assert_allclose(bk.rhat(chains, ops=ops), z["rhat"], rtol=1e-12)
    np.testing.assert_allclose(bk.split_rhat(chains, ops=ops), z["split_rhat"], rtol=1e-12)
    # device matrix [N, C]
    x = torch.from_numpy(np.ascontiguousarray(z["rhat_chains"].T)).to(ops.device)
    np.testing.assert_allclose(bk.rhat(x, ops=ops), z["rhat"], rtol=1e-12)
    np.testing.assert_allclose(bk.split_rhat(x, ops=ops), z["split_rhat"], rtol=1e-12)
    # ragged (rhat.py:163-171 allows it)
    ragged = np.split(z["ragged_flat"], np.cumsum(z["ragged_lens"])[:-1])
    np.testing.assert_allclose(bk.rhat(ragged, ops=ops), z["ragged_rhat"], rtol=1e-12)
    np.testing.assert_allclose(bk.split_rhat(ragged, ops=ops), z["ragged_split_rhat"], rtol=1e-12)
    # rank-normalised R-hat (rhat.py:205-236) incl. the known answers of test/test_rhat.py:113-156
    rk = list(z["rank_chains"])
    np.testing.assert_allclose(bk.rank_normalized_rhat(rk, ops=ops), z["rank_normalized_rhat"], rtol=1e-12)
    xr = torch.from_numpy(np.ascontiguousarray(z["rank_chains"].T)).to(ops.device)
    np.testing.assert_allclose(bk.rank_normalized_rhat(xr, ops=ops), z["rank_normalized_rhat"], rtol=1e-12)
    from bayes_kit_amd.rhat import rank_chains, rank_normalize_chains

    got = rank_normalize_chains(rk, ops=ops)
    np.testing.assert_allclose(np.asarray(got), z["rank_normalized"], rtol=1e-13, atol=0)
    # pooled ranks of draws with MANY ties (small integers), as the reference ranks them (fixture from the reference)
    # The reference ranks with `argsort().argsort()` (rhat.py:51-52), numpy's default UNSTABLE sort: which of several equal
    # draws gets which rank is numpy's implementation detail (the fixture records what this numpy did).  What the reference
    # does pin -- and what is checked: every group of equal values receives the same SET of ranks; within a group this library
    # hands them out in pooled order (stable).  R-hat of such data depends on the order inside the groups only weakly.
    tc = list(z["ties_chains"])
    got = np.concatenate([np.asarray(r) for r in rank_chains(tc, ops=ops)])
    want, vals = z["ties_ranks"].reshape(-1), z["ties_chains"].reshape(-1)
    for v in np.unique(vals):
        np.testing.assert_array_equal(np.sort(got[vals == v]), np.sort(want[vals == v]))
        assert np.all(np.diff(got[vals == v]) > 0)   # stable: pooled order inside a group of ties
    # (R-hat of such heavily tied data DOES depend on the order inside the groups: 1.0028 with this numpy's unstable order,
    # 1.0160 with the pooled order -- earlier chains then hold the lower ranks of every group.  Neither is pinned by the
    # reference; the stable order is the reproducible one.)
    assert abs(float(bk.rank_normalized_rhat(tc, ops=ops)) - 1.0) < 0.05
    assert [list(r) for r in rank_chains([[4.2, 5.7], [7.2, 6.1], [-12.9, 107]], ops=ops)] == [[2, 3], [5, 4], [1, 6]]
    got = rank_normalize_chains([[4.2, 5.7], [7.2, 6.1], [-12.9, 107]], ops=ops)
    np.testing.assert_allclose(got, [[-0.550, -0.087], [0.889, 0.356], [-1.188, 2.225]], atol=2e-2)
    import scipy.stats as st

    want = [[st.norm.ppf((r - 0.325) / (6 - 0.25)) for r in row] for row in [[2, 3], [5, 4], [1, 6]]]
    np.testing.assert_allclose(got, want, rtol=1e-14)
    # streaming moments == two-pass mean / var
    D, C, N = 3, 16, 50
    rng = np.random.default_rng(0)
    draws = rng.normal(size=(N, C, D)) + rng.normal(size=(1, C, 1)) * 0.2
    mom = bk.RunningMoments(D, C, ops=ops)
    for n in range(N):
        mom.update(torch.from_numpy(draws[n]).to(ops.device))
    from oracle import diagnostics as od

    want = [od.rhat([draws[:, c, d] for c in range(C)]) for d in range(D)]
    np.testing.assert_allclose(mom.rhat(), want, rtol=1e-10)
    # draw storage: tracked coordinates + logp, consumed by ess / rhat without reshaping
    rec = bk.DrawRecorder([0, 2], N, C, ops=ops)
    for n in range(N):
        rec.record(torch.from_numpy(draws[n]).to(ops.device), torch.from_numpy(draws[n].sum(axis=1)).to(ops.device))
    assert rec.names() == ["theta[0]", "theta[2]", "logp"] and rec.ess().shape == (3, C)
    np.testing.assert_allclose(rec.rhat()[:2], [want[0], want[2]], rtol=1e-10)
    np.testing.assert_allclose(rec.ess()[1].cpu().numpy(), [od.ess(draws[:, c, 2]) for c in range(C)], rtol=ess_rtol)
    # ESS / IAT / autocorr (ess.py, iat.py, autocorr.py): direct sums vs the reference's FFT
    ar = z["ar_chains"]
    xm = torch.from_numpy(np.ascontiguousarray(ar.T)).to(ops.device)
    np.testing.assert_allclose(bk.ess(xm, ops=ops).cpu().numpy(), z["ar_ess"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.ess_imse(xm, ops=ops).cpu().numpy(), z["ar_ess_imse"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.ess_ipse(xm, ops=ops).cpu().numpy(), z["ar_ess_ipse"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.iat(xm, ops=ops).cpu().numpy(), z["ar_iat"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.iat_ipse(xm, ops=ops).cpu().numpy(), z["ar_iat_ipse"], rtol=ess_rtol)
    ac = bk.autocorr(xm, ops=ops).cpu().numpy()
    np.testing.assert_allclose(ac.T, z["ar_autocorr"], rtol=0, atol=1e-12)
    for i in (0, 7, 15):  # 1-D host inputs, as the reference is called
        np.testing.assert_allclose(bk.ess(ar[i], ops=ops), z["ar_ess"][i], rtol=ess_rtol)
        np.testing.assert_allclose(bk.autocorr(ar[i], ops=ops), z["ar_autocorr"][i], rtol=0, atol=1e-12)
    short = np.split(z["short_flat"], np.cumsum(z["short_lens"])[:-1])
    np.testing.assert_allclose([bk.ess(c, ops=ops) for c in short], z["short_ess"], rtol=ess_rtol)
    np.testing.assert_allclose(np.concatenate([bk.autocorr(c, ops=ops) for c in short]), z["short_autocorr_flat"],
                               rtol=0, atol=1e-12)
    # long chains (20,000 draws >= FFT_MIN_DRAWS: the library's own FFT), 1-D as the reference is called and as one [N, C] array
    from tests.helpers import long_ar_chains

    lc = long_ar_chains(z["long_seed"], z["long_n"], z["long_phi"])
    xl = torch.from_numpy(np.ascontiguousarray(np.stack(lc).T)).to(ops.device)
    np.testing.assert_allclose(bk.ess(xl, ops=ops).cpu().numpy(), z["long_ess"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.ess_ipse(xl, ops=ops).cpu().numpy(), z["long_ess_ipse"], rtol=ess_rtol)
    np.testing.assert_allclose(bk.iat(xl, ops=ops).cpu().numpy(), z["long_iat"], rtol=ess_rtol)
    acl = bk.autocorr(xl, ops=ops).cpu().numpy()
    np.testing.assert_allclose(acl[:64].T, z["long_autocorr_head"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(acl[-8:].T, z["long_autocorr_tail"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(bk.ess(lc[2], ops=ops), z["long_ess"][2], rtol=ess_rtol)
    # known answer held by the reference's test (test/test_autocorr.py:10-14)
    np.testing.assert_allclose(bk.autocorr([1, 0, 0, 0], ops=ops), [1.0, -0.083, -0.167, -0.25], atol=0.001)
    # error behaviour (rhat.py:159-162, ess.py:67-68, iat.py, autocorr.py:23-24)
    with pytest.raises(ValueError):
        bk.rhat([[1.0, 2.0]], ops=ops)
    with pytest.raises(ValueError):
        bk.rhat([[1.0, 2.0], [1.0]], ops=ops)
    for f in (bk.ess, bk.ess_imse, bk.ess_ipse, bk.iat, bk.iat_imse, bk.iat_ipse):
        with pytest.raises(ValueError):
            f([1.0, 2.0, 3.0], ops=ops)
    with pytest.raises(ValueError):
        bk.autocorr([1.0], ops=ops)


def check_checkpoint_of_sampler_and_diagnostics(ops, tmpdir, chains=40, D=12, draws=40, at=20):
    """{sampler, Welford moments, tracked series, chunked draw store} checkpointed after
    `at` draws and restored into FRESH objects give, after the remaining draws, bit for bit what an
    uninterrupted run gives (moments, R-hat, ESS, every stored draw)."""
    import io
    import os

    import bayes_kit_amd as bk

    lam = np.logspace(0, 1, D)

    def fresh(path, resume=False):
        s = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.15, 4, chains=chains, seed=77, ops=ops)
        return (s, bk.RunningMoments(D, chains, ops=ops), bk.DrawRecorder([0, D - 1], draws, chains, ops=ops),
                bk.DrawStore.create(path, D, chains, chunk=8, ops=ops, resume=resume))

    def step(objs, n):
        s, mom, rec, store = objs
        for _ in range(n):
            th, lp = s.sample()
            mom.update(th)
            rec.record(th, lp)
            store.append(th)

    def summary(objs):
        s, mom, rec, store = objs
        store.close()
        rd = bk.DrawStore.open(store.path, ops=ops)
        assert rd.draws == draws
        return dict(mean=mom.mean.cpu().numpy().copy(), m2=mom.m2.cpu().numpy().copy(), rhat=mom.rhat(),
                    series=rec.series[:, : rec.n].cpu().numpy().copy(), ess=rec.ess().cpu().numpy(),
                    store0=rd.series(0).cpu().numpy(), store5=rd.series(D // 2).cpu().numpy(), store_rhat=rd.rhat([1, D - 2]),
                    store_ess=rd.ess([D - 1]).cpu().numpy(), theta=np.asarray(s._theta.cpu()).copy(), rng=s.rng_state())

    a = fresh(os.path.join(tmpdir, "a"))
    step(a, draws)
    want = summary(a)

    b = fresh(os.path.join(tmpdir, "b"))
    step(b, at)
    buf = io.BytesIO()
    torch.save({"sampler": b[0].state_dict(), "moments": b[1].state_dict(), "recorder": b[2].state_dict(),
                "store": b[3].state_dict()}, buf)
    step(b, 13)  # the interrupted run went on for a while (one more chunk file reached the disk) and died
    del b
    c = fresh(os.path.join(tmpdir, "b"), resume=True)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    c[0].load_state_dict(ck["sampler"])
    c[1].load_state_dict(ck["moments"])
    c[2].load_state_dict(ck["recorder"])
    c[3].load_state_dict(ck["store"])
    step(c, draws - at)
    got = summary(c)
    for k in want:
        assert np.array_equal(want[k], got[k]), k
    # the stored series of a coordinate is what ess / rhat of the recorder saw
    assert np.array_equal(want["store0"], want["series"][0])
    # the chunk files the interrupted run wrote after the checkpoint were set aside, not deleted
    assert any(f.startswith("superseded_0_chunk_") for f in os.listdir(os.path.join(tmpdir, "b")))
    check_square_draws_and_reader(ops, os.path.join(tmpdir, "sq"))


def check_square_draws_and_reader(ops, path):
    """C == D: the (C, D) view sample() returns and a [D, C] buffer have the same shape; the store and the
    moments tell them apart by strides (or are told).  A reader never allocates the staging buffer."""
    import bayes_kit_amd as bk

    n = 6
    lam = np.logspace(0, 0.5, n)
    s = bk.HMCDiag(bk.DiagGaussian(lam, ops=ops), 0.2, 3, chains=n, seed=5, ops=ops)
    store = bk.DrawStore.create(path, n, n, chunk=4, ops=ops)
    assert store._buf_t is None  # nothing staged yet
    mom_view, mom_buf, mom_told = (bk.RunningMoments(n, n, ops=ops) for _ in range(3))
    kept = []
    for i in range(6):
        th, _ = s.sample()                       # (C, D) view, chain stride 1
        assert th.stride(0) == 1 or n == 1
        kept.append(np.array(th.cpu().numpy()))
        store.append(th) if i % 2 == 0 else store.append(th.t().contiguous().t(), layout="cd")
        mom_view.update(th)
        mom_buf.update(th.t())                   # the [D, C] buffer itself
        mom_told.update(th.contiguous(), layout="cd")  # a row-major copy: strides say nothing, the caller does
        with pytest.raises(ValueError):
            mom_told.update(torch.zeros((n, n), dtype=torch.float64, device=ops.device)[:, :1].expand(n, n))
    store.close()
    rd = bk.DrawStore.open(path, ops=ops)
    for d in (0, n - 1):
        want = np.stack([k[:, d] for k in kept])   # [N, C]: coordinate d of every chain
        assert np.array_equal(rd.series(d).cpu().numpy(), want), d
    assert rd._buf_t is None  # the reader read 6 draws without a staging buffer
    assert torch.equal(mom_view.mean, mom_buf.mean) and torch.equal(mom_view.mean, mom_told.mean)
    assert np.array_equal(mom_view.mean.cpu().numpy(), np.mean(np.stack(kept), axis=0).T) or np.allclose(
        mom_view.mean.cpu().numpy(), np.mean(np.stack(kept), axis=0).T, rtol=1e-13)
