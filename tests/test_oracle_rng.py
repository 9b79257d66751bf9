"""oracle/rng.py vs numpy.random itself (the un-vendored dependency the reference uses)."""
import math

import numpy as np

from oracle.rng import PhiloxStream, log1p_fdlibm, philox4x64_10, ziggurat_tables


def test_tables_check_values():
    ki, wi, fi = ziggurat_tables()
    assert ki[0] == 0x000EF33D8025EF6A and ki[1] == 0
    assert wi[0] == 8.683627060801306e-16 and wi[255] == 8.113849337656484e-16
    assert fi[255] == 0.001260285930498598 and fi[0] == 1.0


def test_philox_raw_words_incl_carry():
    for key in ([1, 2], [2**64 - 1, 0x9E3779B97F4A7C15]):
        raw = np.random.Philox(key=key).random_raw(41)
        s = PhiloxStream(*key)
        assert [s.next_u64() for _ in range(41)] == [int(v) for v in raw]
    # carry out of the low counter word
    bg = np.random.Philox(key=[7, 9], counter=np.array([2**64 - 1, 2**64 - 1, 5, 0], dtype=np.uint64))
    s = PhiloxStream(7, 9)
    s.counter = [2**64 - 1, 2**64 - 1, 5, 0]
    assert [s.next_u64() for _ in range(9)] == [int(v) for v in bg.random_raw(9)]
    assert s.counter == [2, 0, 6, 0]
    assert [int(v) for v in bg.state["state"]["counter"]] == s.counter
    assert philox4x64_10([0, 0, 0, 0], [0, 0]) != [0, 0, 0, 0]


def test_normals_uniforms_and_state_bit_exact():
    key = [20240, 17]
    g = np.random.Generator(np.random.Philox(key=key))
    s = PhiloxStream(*key, log1p=log1p_fdlibm)
    n = 150_000  # ~37 tail draws, ~2200 wedge draws
    ref = g.normal(size=n)
    mine = np.array([s.normal() for _ in range(n)])
    assert np.array_equal(ref.view(np.uint64), mine.view(np.uint64))
    # interleaved normal(size=7) / uniform() as HMC consumes them (hmc.py:56,60)
    for _ in range(50):
        a = g.normal(size=7)
        b = np.array([s.normal() for _ in range(7)])
        assert np.array_equal(a, b)
        assert g.uniform() == s.uniform()
    # loc/scale form used by DRGHMC (drghmc.py:360-364): loc + scale*z
    loc = np.linspace(-1, 1, 5)
    a = g.normal(loc=loc, scale=0.3, size=5)
    b = np.array([loc[i] + 0.3 * s.normal() for i in range(5)])
    assert np.array_equal(a, b)
    st = g.bit_generator.state
    assert [int(v) for v in st["state"]["counter"]] == s.counter
    assert [int(v) for v in st["buffer"]] == s.buffer
    assert st["buffer_pos"] == s.buffer_pos


def test_log1p_restatement_matches_libm():
    rng = np.random.default_rng(3)
    xs = np.concatenate([
        -rng.random(60000), -rng.random(20000) * 1e-3, -(1 - rng.random(20000) * 1e-6),
        -rng.random(10000) * 2.0**-30, -rng.random(2000) * 2.0**-55, [0.0, -0.5, -0.2929, -0.29290001]])
    for x in xs:
        assert log1p_fdlibm(float(x)) == math.log1p(float(x)), x


# ---- the legacy global stream of bayes_kit/smc.py:73,81,85 (numpy.random.RandomState) -------------------------------
def test_legacy_mt19937_stream_matches_numpy_randomstate():
    from oracle.rng import LegacyStream

    for seed in (0, 1, 20245, 2**32 - 1):
        rs, o = np.random.RandomState(seed), LegacyStream(seed)
        assert np.array_equal(rs.get_state()[1], np.array(o.mt, dtype=np.uint32))
        for k in range(400):  # D normals then one uniform, D odd and even: the cached gauss crosses the uniforms
            D = 1 + k % 5
            loc = np.arange(D) * 0.1
            assert np.array_equal(rs.normal(loc=loc, scale=0.3), o.normal(loc, 0.3)), (seed, k)
            assert rs.uniform() == o.uniform()
        assert np.array_equal(rs.random_sample(700), o.choice_uniforms(700))  # crosses a 624-word refill
        st = rs.get_state(legacy=False)
        assert (st["has_gauss"], st["gauss"], st["state"]["pos"]) == (o.has_gauss, o.gauss, o.pos)
        assert np.array_equal(st["state"]["key"], o.state()["key"])


def test_numpy_sum_and_choice_restatements():
    from oracle.rng import legacy_choice, numpy_pairwise_sum

    g = np.random.default_rng(1)
    for n in list(range(1, 140)) + [255, 256, 257, 511, 1000, 2048, 4097, 8191, 8192, 8193, 10000, 16385, 20000, 70001]:
        w = np.exp(g.normal(size=n) * 3)
        s = np.sum(w)
        assert s == numpy_pairwise_sum(w), n
        p = w / s
        rs = np.random.RandomState(n)
        st = rs.get_state()
        idx = rs.choice(n, size=n, replace=True, p=p)
        rs.set_state(st)
        assert np.array_equal(idx, legacy_choice(p, rs.random_sample(n))), n


def test_exact_fma_and_the_library_exp_restatement():
    """oracle.rng.fma_exact is the correctly rounded a*b+c (checked where the answer is known exactly); oracle.rng.exp_bk --
    the HIP library's bk_exp restated -- stays within one ulp of the host libm's exp and equals it for most arguments."""
    import math

    from oracle.rng import exp_bk, fma_exact

    rng = np.random.default_rng(8)
    for a, b in rng.normal(size=(2000, 2)):
        a, b = float(a), float(b)
        p = a * b
        e = fma_exact(a, b, -p)                     # the rounding error of the product, exactly representable
        na, da = a.as_integer_ratio()
        nb, db = b.as_integer_ratio()
        npn, dp = p.as_integer_ratio()
        ne, de = e.as_integer_ratio()
        assert na * nb * dp * de == (npn * de + ne * dp) * da * db          # a * b == p + e exactly
        assert fma_exact(a, b, 0.0) == p and fma_exact(a, 1.0, b) == a + b
    assert fma_exact(1.0 + 2.0 ** -52, 1.0 + 2.0 ** -52, -1.0) == 2.0 ** -51 + 2.0 ** -104   # (a * b rounds the last term away)
    assert fma_exact(2.0 ** -600, 2.0 ** -600, 0.0) == 0.0 and fma_exact(3.0, 2.0 ** -1074, 2.0 ** -1074) == 4 * 2.0 ** -1074
    assert math.copysign(1.0, fma_exact(-1.0, 0.0, -0.0)) == -1.0 and math.copysign(1.0, fma_exact(1.0, 0.0, -0.0)) == 1.0
    xs = np.concatenate([rng.normal(size=4000) * 4.0, rng.uniform(-700.0, 700.0, 2000), rng.uniform(-0.35, 0.35, 1000)])
    same = 0
    for x in xs:
        x = float(x)
        got, ref = exp_bk(x), math.exp(x)
        assert abs(got - ref) <= abs(np.nextafter(ref, np.inf) - ref), x
        same += got == ref
    assert same >= 0.85 * len(xs)
    assert exp_bk(0.0) == 1.0 and exp_bk(-0.0) == 1.0 and exp_bk(float("inf")) == float("inf") and exp_bk(float("-inf")) == 0.0
    assert exp_bk(709.782712893384) == 1.7976931348622732e308 and exp_bk(709.7827128933841) == float("inf")
    assert exp_bk(-745.13) == 5e-324 and exp_bk(-745.14) == 0.0 and math.isnan(exp_bk(float("nan")))
