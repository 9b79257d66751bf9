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
