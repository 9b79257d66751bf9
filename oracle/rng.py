"""Pure-Python restatement of the random stream the reference consumes (TEST INFRASTRUCTURE).

The reference draws everything from ``numpy.random.Generator`` (``bayes_kit/hmc.py:23,56,60``,
``mala.py:25,44,55-61``, ``drghmc.py:71,77,360-364,370,378``).  ``numpy.random`` is a
third-party dependency that is not vendored in ``/root/reference``; the pinned version is
the image's numpy 2.2.6.  Its published algorithms, restated here on Python ints/floats:

* bit generator: Philox4x64-10 (Random123; numpy ``random/src/philox/philox.h``): 256-bit
  counter incremented BEFORE each block, four 64-bit outputs consumed in order;
* ``uniform()``: ``(next_u64 >> 11) * 2**-53``;
* ``normal()``: the 256-layer ziggurat of numpy ``random/src/distributions/distributions.c``
  (``random_standard_normal``), tables extracted from numpy by
  ``tests/golden/make_ziggurat_tables.py``;
* the ziggurat tail uses the host libm's ``log1p``; ``log1p_fdlibm`` restates Sun fdlibm 5.3 ``s_log1p.c`` (argument
  reduction, thresholds, constants; notice in THIRD_PARTY.md) with the degree-7 polynomial summed in pairs -- the
  order this image's libm uses, which is what decides the last bit -- so the device can reproduce tail draws
  bit-for-bit without libm; ``tests/test_oracle_rng.py`` checks it against ``math.log1p``.

``tests/test_oracle_rng.py`` checks each piece against numpy itself (raw words, doubles,
normals, final ``bit_generator.state``), so this file is pinned by the dependency's own
output.
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np

M64 = (1 << 64) - 1
PHILOX_M0 = 0xD2E7470EE14C6C93
PHILOX_M1 = 0xCA5A826395121157
PHILOX_W0 = 0x9E3779B97F4A7C15
PHILOX_W1 = 0xBB67AE8584CAA73B

ZIG_R = 3.6541528853610087963519472518
ZIG_INV_R = 0.27366123732975827203338247596

_TABLES = None


def ziggurat_tables():
    """(ki uint64[256], wi float64[256], fi float64[256]) from the committed fixture."""
    global _TABLES
    if _TABLES is None:
        path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ziggurat_tables.npz")
        z = np.load(path)
        _TABLES = (
            [int(v) for v in z["ki"]],
            [float(v) for v in z["wi"]],
            [float(v) for v in z["fi"]],
        )
    return _TABLES


def philox4x64_10(ctr, key):
    """One Philox4x64-10 block: ctr (4 ints), key (2 ints) -> 4 ints."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for rnd in range(10):
        if rnd:
            k0 = (k0 + PHILOX_W0) & M64
            k1 = (k1 + PHILOX_W1) & M64
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        c0, c1, c2, c3 = (
            ((p1 >> 64) ^ c1 ^ k0) & M64,
            p1 & M64,
            ((p0 >> 64) ^ c3 ^ k1) & M64,
            p0 & M64,
        )
    return [c0, c1, c2, c3]


def _hi_word(x: float) -> int:
    """Signed high 32 bits of the IEEE-754 pattern of x."""
    return struct.unpack("<i", struct.pack("<d", x)[4:])[0]


def _with_hi_word(x: float, hi: int) -> float:
    lo = struct.pack("<d", x)[:4]
    return struct.unpack("<d", lo + struct.pack("<i", hi))[0]


_LN2_HI = 6.93147180369123816490e-01
_LN2_LO = 1.90821492927058770002e-10
_LP = (
    0.0,
    6.666666666666735130e-01,
    3.999999999940941908e-01,
    2.857142874366239149e-01,
    2.222219843214978396e-01,
    1.818357216161805012e-01,
    1.531383769920937332e-01,
    1.479819860511658591e-01,
)


def log1p_fdlibm(x: float) -> float:
    """fdlibm ``log1p`` (polynomial summed in pairs, as the host libm does) for finite x > -1: the only inputs the
    ziggurat produces.  Equal to ``math.log1p`` on this image bit for bit (tested)."""
    hx = _hi_word(x)
    ax = hx & 0x7FFFFFFF
    k = 1
    f = 0.0
    hu = 0
    c = 0.0
    if hx < 0x3FDA827A:  # x < 0.41422
        if ax >= 0x3FF00000:
            return -math.inf if x == -1.0 else math.nan
        if ax < 0x3E200000:  # |x| < 2**-29
            if ax < 0x3C900000:
                return x
            return x - x * x * 0.5
        if hx > 0 or hx <= -0x402D413D:  # (int32)0xbfd2bec3 : -0.2929 < x < 0.41422
            k, f, hu = 0, x, 1
    if k != 0:
        u = 1.0 + x
        hu = _hi_word(u)
        k = (hu >> 20) - 1023
        c = (1.0 - (u - x)) if k > 0 else (x - (u - 1.0))
        c /= u
        hu &= 0x000FFFFF
        if hu < 0x6A09E:
            u = _with_hi_word(u, hu | 0x3FF00000)
        else:
            k += 1
            u = _with_hi_word(u, hu | 0x3FE00000)
            hu = (0x00100000 - hu) >> 2
        f = u - 1.0
    hfsq = 0.5 * f * f
    if hu == 0:  # |f| < 2**-20
        if f == 0.0:
            if k == 0:
                return 0.0
            c += k * _LN2_LO
            return k * _LN2_HI + c
        R = hfsq * (1.0 - 0.66666666666666666 * f)
        if k == 0:
            return f - R
        return k * _LN2_HI - ((R - (k * _LN2_LO + c)) - f)
    s = f / (2.0 + f)
    z = s * s
    R1 = z * _LP[1]
    z2 = z * z
    R2 = _LP[2] + z * _LP[3]
    z4 = z2 * z2
    R3 = _LP[4] + z * _LP[5]
    z6 = z4 * z2
    R4 = _LP[6] + z * _LP[7]
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4
    if k == 0:
        return f - (hfsq - s * (hfsq + R))
    return k * _LN2_HI - ((hfsq - (s * (hfsq + R) + (k * _LN2_LO + c))) - f)


class PhiloxStream:
    """numpy.random.Philox(key=[k0, k1]) + Generator.uniform()/normal(), one value at a time."""

    def __init__(self, key0: int, key1: int, log1p=math.log1p):
        self.key = [key0 & M64, key1 & M64]
        self.counter = [0, 0, 0, 0]
        self.buffer = [0, 0, 0, 0]
        self.buffer_pos = 4  # empty
        self._log1p = log1p

    def next_u64(self) -> int:
        if self.buffer_pos < 4:
            v = self.buffer[self.buffer_pos]
            self.buffer_pos += 1
            return v
        for i in range(4):  # 256-bit increment with carry, BEFORE the block
            self.counter[i] = (self.counter[i] + 1) & M64
            if self.counter[i] != 0:
                break
        self.buffer = philox4x64_10(self.counter, self.key)
        self.buffer_pos = 1
        return self.buffer[0]

    def next_double(self) -> float:
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    uniform = next_double

    def normal(self) -> float:
        ki, wi, fi = ziggurat_tables()
        while True:
            r = self.next_u64()
            idx = r & 0xFF
            r >>= 8
            sign = r & 1
            rabs = (r >> 1) & 0x000FFFFFFFFFFFFF
            x = rabs * wi[idx]
            if sign:
                x = -x
            if rabs < ki[idx]:
                return x
            if idx == 0:
                while True:
                    xx = -ZIG_INV_R * self._log1p(-self.next_double())
                    yy = -self._log1p(-self.next_double())
                    if yy + yy > xx * xx:
                        return -(ZIG_R + xx) if (rabs >> 8) & 1 else ZIG_R + xx
            else:
                if (fi[idx - 1] - fi[idx]) * self.next_double() + fi[idx] < math.exp(-0.5 * x * x):
                    return x

    def state(self):
        return {
            "key": list(self.key),
            "counter": list(self.counter),
            "buffer": list(self.buffer),
            "buffer_pos": self.buffer_pos,
        }


class LegacyStream:
    """The process-global ``numpy.random`` stream the reference's SMC draws from (``bayes_kit/smc.py:73,81,85``):
    ``numpy.random.RandomState`` = MT19937 (Matsumoto & Nishimura 1998; numpy ``random/src/mt19937/mt19937.c``) with
    the LEGACY distributions of ``random/src/legacy/legacy-distributions.c`` -- frozen by NumPy's compatibility policy
    (NEP 19), restated here on Python ints / floats:

    * ``seed(s)``, 0 <= s < 2**32: Knuth's recurrence ``mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i``;
    * a 32-bit word: the standard tempering of the 624-word state, regenerated 624 words at a time;
    * ``random_sample()`` / ``uniform()``: ``((a >> 5) * 2**26 + (b >> 6)) / 2**53`` from two words;
    * ``normal()``: Marsaglia's polar method, ``x1, x2 = 2 u - 1`` until ``0 < r2 < 1``, ``f = sqrt(-2 log(r2) / r2)``,
      returns ``f * x2`` and KEEPS ``f * x1`` for the next call (the cached value survives ``uniform()`` calls in
      between: with D odd the second half of a pair crosses from one particle to the next);
      ``normal(loc, scale) = loc + scale * gauss``;
    * ``choice(n, size, replace=True, p)``: ``cdf = p.cumsum(); cdf /= cdf[-1]``, ``size`` doubles,
      ``cdf.searchsorted(u, side="right")``.

    ``tests/test_oracle_rng.py`` pins every piece against ``numpy.random.RandomState`` itself."""

    N, M_ = 624, 397

    def __init__(self, seed: int):
        seed = int(seed)
        if not 0 <= seed < (1 << 32):
            raise ValueError("legacy integer seed must fit 32 bits")
        mt = [0] * self.N
        mt[0] = seed
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.pos = mt, self.N
        self.has_gauss, self.gauss = 0, 0.0

    def _regenerate(self):
        mt, N, M = self.mt, self.N, self.M_
        for k in range(N):
            y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
            mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        self.pos = 0

    def next_u32(self) -> int:
        if self.pos == self.N:
            self._regenerate()
        y = self.mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y

    def random_sample(self) -> float:
        a = self.next_u32() >> 5
        b = self.next_u32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    uniform = random_sample

    def standard_normal(self) -> float:
        if self.has_gauss:
            self.has_gauss, g = 0, self.gauss
            self.gauss = 0.0
            return g
        while True:
            x1 = 2.0 * self.random_sample() - 1.0
            x2 = 2.0 * self.random_sample() - 1.0
            r2 = x1 * x1 + x2 * x2
            if r2 < 1.0 and r2 != 0.0:
                break
        f = math.sqrt(-2.0 * math.log(r2) / r2)
        self.gauss, self.has_gauss = f * x1, 1
        return f * x2

    def normal(self, loc, scale) -> np.ndarray:
        """``np.random.normal(loc=array, scale=float)``: one gauss per element, ``loc + scale * gauss``."""
        loc = np.atleast_1d(np.asarray(loc, dtype=np.float64))
        return np.array([loc[i] + scale * self.standard_normal() for i in range(loc.shape[0])])

    def choice_uniforms(self, m: int) -> np.ndarray:
        return np.array([self.random_sample() for _ in range(m)])

    def state(self):
        """As ``RandomState.get_state(legacy=False)`` reports it."""
        return dict(key=np.array(self.mt, dtype=np.uint32), pos=self.pos, has_gauss=self.has_gauss, gauss=self.gauss)


def numpy_pairwise_sum(a) -> float:
    """``np.sum`` of a contiguous float64 vector, restated: numpy's pairwise summation (``DOUBLE_pairwise_sum`` in
    ``numpy/_core/src/umath/loops_utils.h.src``): fewer than 8 values -> a plain loop; up to 128 -> eight running
    sums over strides of 8 combined ``((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))`` plus the remainder in order; beyond that
    the vector is split at ``n/2`` rounded down to a multiple of 8 and the halves are summed recursively.  The
    reduction machinery hands the inner loop at most 8,192 values at a time (its buffer size) and adds the pieces' sums
    up in order -- probed against ``np.sum`` at lengths around and beyond 8,192 (``tests/test_oracle_rng.py``)."""
    a = [float(v) for v in a]

    def pw(lo, n):
        if n < 8:
            res = 0.0
            for i in range(n):
                res += a[lo + i]
            return res
        if n <= 128:
            r = a[lo:lo + 8]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] += a[lo + i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res += a[lo + i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(lo, n2) + pw(lo + n2, n - n2)

    acc = pw(0, min(len(a), 8192))
    for lo in range(8192, len(a), 8192):
        acc = acc + pw(lo, min(8192, len(a) - lo))
    return acc


def legacy_choice(p, u) -> np.ndarray:
    """Indices ``RandomState.choice(len(p), size=len(u), replace=True, p=p)`` returns when its uniforms are ``u``."""
    cdf = np.empty(len(p))
    acc = 0.0
    for i, v in enumerate(p):
        acc = acc + float(v) if i else float(v)
        cdf[i] = acc
    last = cdf[-1]
    cdf = np.array([c / last for c in cdf])
    out = np.empty(len(u), dtype=np.int64)
    for j, x in enumerate(u):
        lo, hi = 0, len(cdf)
        while lo < hi:  # first index with cdf > x  (side="right")
            mid = (lo + hi) // 2
            if cdf[mid] <= x:
                lo = mid + 1
            else:
                hi = mid
        out[j] = lo
    return out


def fma_exact(a: float, b: float, c: float) -> float:
    """IEEE 754-2008 fusedMultiplyAdd on finite doubles: a * b + c evaluated exactly (integer ratios; every double is
    one) and rounded ONCE to nearest even -- CPython's int / int is correctly rounded.  (``math.fma`` is Python >= 3.13.)"""
    na, da = float(a).as_integer_ratio()
    nb, db = float(b).as_integer_ratio()
    nc, dc = float(c).as_integer_ratio()
    num, den = na * nb * dc + nc * da * db, da * db * dc
    if num == 0:  # IEEE: the sign of an exact zero sum (round to nearest: +0 unless both addends are -0)
        prod_neg = (math.copysign(1.0, a) * math.copysign(1.0, b)) < 0
        return -0.0 if (prod_neg and math.copysign(1.0, c) < 0) else 0.0
    return num / den


_EXP_BK_C = tuple(float.fromhex(h) for h in (
    "0x1.0000000000000p-1", "0x1.5555555555555p-3", "0x1.5555555555555p-5", "0x1.1111111111111p-7",
    "0x1.6c16c16c16c17p-10", "0x1.a01a01a01a01ap-13", "0x1.a01a01a01a01ap-16", "0x1.71de3a556c734p-19",
    "0x1.27e4fb7789f5cp-22", "0x1.ae64567f544e4p-26", "0x1.1eed8eff8d898p-29", "0x1.6124613a86d09p-33"))  # 1/2! .. 1/13!


def exp_bk(x: float) -> float:
    """The HIP library's ``bk_exp`` (``include/bkhip_math.h``) restated operation for operation on Python floats --
    x clamped to [-746, 710]; k = rint(x log2 e); r = fma(-k, LN2_HI, x); r = fma(-k, LN2_LO, r); Horner with fma over 1/13! .. 1/2!;
    p = fma(r r, q, r) + 1; ldexp(p, k) -- so that an oracle-side density with an exp() inside is evaluated with the
    device's own double (``oracle.models.FunnelCanonical``).  ``tests/test_abi.py`` checks it against the library's host
    build of the header; the GPU tests check the device build against it."""
    log2e, ln2_hi, ln2_lo = float.fromhex("0x1.71547652b82fep+0"), float.fromhex("0x1.62e42fefa39efp-1"), \
        float.fromhex("0x1.abc9e3b39803fp-56")
    x = float(x)
    if x != x:
        return x + x
    x = min(max(x, -746.0), 710.0)  # beyond these ldexp returns 0 / inf by itself
    k = float(round(x * log2e))  # (round() of a float: to nearest, ties to even, as rint in the default mode)
    r = fma_exact(-k, ln2_hi, x)
    r = fma_exact(-k, ln2_lo, r)
    q = _EXP_BK_C[11]
    for n in range(10, -1, -1):
        q = fma_exact(q, r, _EXP_BK_C[n])
    p = fma_exact(r * r, q, r)
    p = p + 1.0
    try:
        return math.ldexp(p, int(k))
    except OverflowError:  # (C's ldexp returns +inf)
        return float("inf")
