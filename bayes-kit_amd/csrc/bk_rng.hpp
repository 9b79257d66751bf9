// Per-chain random streams for the many-chain samplers: bit-for-bit the stream that
// numpy.random.Generator hands the reference samplers (bayes_kit/hmc.py:23,56,60;
// mala.py:25,44; drghmc.py:71,77,360-364,370,378).
//
//   * bit generators: Philox4x64-10 (np.random.Philox(key=[k0,k1])) and PCG64
//     (np.random.default_rng(int) -> PCG64 XSL-RR 128/64), state held per chain in a
//     caller-owned SoA table  state[BK_RNG_WORDS][ld]  of u64 (one chain per column);
//   * double in [0,1): (u64 >> 11) * 2^-53;
//   * standard normal: NumPy's 256-layer ziggurat with NumPy's own tables
//     (ziggurat_tables.inc), including the tail branch, which needs the host libm's log1p
//     reproduced exactly (bk_log1p below: fdlibm's) for the draws to be bit-identical.
//
// Everything here is __host__ __device__ so the exact source the GPU runs can also be
// exercised on the host by the library's bk_host_* self-test hooks (tests only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define BK_RNG_WORDS 11  // philox: key[2] ctr[4] buf[4] pos ; pcg64: state_hi state_lo inc_hi inc_lo
#define BK_RNG_PHILOX 0
#define BK_RNG_PCG64 1

#define BK_HD __host__ __device__ __forceinline__

namespace bk {

BK_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// hi and lo halves of the 128-bit product from the SAME four 32x32->64 partial products (the separate
// __umul64hi(a, b) and a * b the compiler is given otherwise cost two extra quarter-rate 32-bit
// multiplies per product: 40 of 120 multiply instructions per Philox block)
BK_HD void mulhilo64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
  const uint64_t p00 = (uint64_t)al * bl;
  const uint64_t p01 = (uint64_t)al * bh + (p00 >> 32);          // <= 2^64 - 2^33 + ... : no overflow
  const uint64_t p10 = (uint64_t)ah * bl + (uint32_t)p01;        // low word of p01 carried in
  hi = (uint64_t)ah * bh + (p01 >> 32) + (p10 >> 32);
  lo = (p10 << 32) | (uint32_t)p00;
#else
  unsigned __int128 p = (unsigned __int128)a * (unsigned __int128)b;
  hi = (uint64_t)(p >> 64);
  lo = (uint64_t)p;
#endif
}

// a ^ b ^ c: one V_BITOP3_B32 per 32-bit half on gfx950 (the compiler emits two v_xor_b32 per half otherwise:
// 80 of the ~260 vector instructions of a Philox block)
BK_HD uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
  const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
  return ((uint64_t)hi << 32) | lo;
#else
  return a ^ b ^ c;
#endif
}

BK_HD double u64_as_double(uint64_t u) {
  union { uint64_t u; double d; } x;
  x.u = u;
  return x.d;
}
BK_HD uint64_t double_as_u64(double d) {
  union { uint64_t u; double d; } x;
  x.d = d;
  return x.u;
}

// ---- Philox4x64-10 -----------------------------------------------------------------
// Stored state (numpy's): key, counter `c` of the block held in `b`, read position `pos`.
// Device-side lookahead: up to three further blocks (n1 = counter c+1, n2 = c+2, n3 = c+3)
// generated AHEAD of need, two at a time with their rounds interleaved.  A Philox block costs ~20 full 64x64->128 multiplies; lanes of a wavefront
// consume their streams at slightly different rates (ziggurat rejections), so without
// lookahead every call would make the whole wave wait for the few lanes that need a new
// block.  Kernels call top_up() at a wave-uniform point: when ANY lane is short, ALL lanes
// with room generate together.  The lookahead is a cache only: it is never stored, and
// store() writes exactly the (counter, buffer, pos) numpy would hold.
struct Philox {
  uint64_t key0, key1;
  uint64_t c0, c1, c2, c3;   // counter of the block in b (numpy's counter)
  uint64_t b0, b1, b2, b3;   // that block, complete (numpy's buffer)
  uint64_t q0, q1, q2, q3;   // its unread words, q0 next (a shift queue: no dynamic indexing,
  uint32_t rem;              //   which the compiler would otherwise spill to scratch); numpy pos = 4 - rem
  uint64_t n10, n11, n12, n13, n20, n21, n22, n23, n30, n31, n32, n33;
  uint32_t cnt;              // lookahead blocks held (0..3)

  BK_HD void block_at(uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3, uint64_t& o0, uint64_t& o1,
                      uint64_t& o2, uint64_t& o3) const {
    const uint64_t M0 = 0xD2E7470EE14C6C93ULL, M1 = 0xCA5A826395121157ULL;
    const uint64_t W0 = 0x9E3779B97F4A7C15ULL, W1 = 0xBB67AE8584CAA73BULL;
    uint64_t k0 = key0, k1 = key1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      if (r) { k0 += W0; k1 += W1; }
      uint64_t hi0, lo0, hi1, lo1;
      mulhilo64(M0, x0, hi0, lo0);
      mulhilo64(M1, x2, hi1, lo1);
      uint64_t y0 = xor3_64(hi1, x1, k0), y2 = xor3_64(hi0, x3, k1);
      x0 = y0; x1 = lo1; x2 = y2; x3 = lo0;
    }
    o0 = x0; o1 = x1; o2 = x2; o3 = x3;
  }

  // The same block with the ten round keys read from a table (rk[2 r], rk[2 r + 1] = key0 + r W0, key1 + r W1):
  // kernels that generate many blocks of ONE stream keep the table in LDS instead of 40 registers or 36 additions
  // per block.
  BK_HD static void block_with_round_keys(const uint64_t* rk, uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3,
                                          uint64_t& o0, uint64_t& o1, uint64_t& o2, uint64_t& o3) {
    const uint64_t M0 = 0xD2E7470EE14C6C93ULL, M1 = 0xCA5A826395121157ULL;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint64_t hi0, lo0, hi1, lo1;
      mulhilo64(M0, x0, hi0, lo0);
      mulhilo64(M1, x2, hi1, lo1);
      uint64_t y0 = xor3_64(hi1, x1, rk[2 * r]), y2 = xor3_64(hi0, x3, rk[2 * r + 1]);
      x0 = y0; x1 = lo1; x2 = y2; x3 = lo0;
    }
    o0 = x0; o1 = x1; o2 = x2; o3 = x3;
  }

  BK_HD uint64_t next() {
    if (rem == 0) {
      // the 256-bit counter is incremented BEFORE the block is generated
      if (++c0 == 0) { if (++c1 == 0) { if (++c2 == 0) { ++c3; } } }
      if (cnt > 0) {
        b0 = n10; b1 = n11; b2 = n12; b3 = n13;
        n10 = n20; n11 = n21; n12 = n22; n13 = n23;
        n20 = n30; n21 = n31; n22 = n32; n23 = n33;
        --cnt;
      } else {
        block_at(c0, c1, c2, c3, b0, b1, b2, b3);
      }
      q0 = b0; q1 = b1; q2 = b2; q3 = b3;
      rem = 4;
    }
    uint64_t v = q0;
    q0 = q1; q1 = q2; q2 = q3;
    --rem;
    return v;
  }

  // push the word just returned by next() back (the caller will consume it again)
  BK_HD void unget(uint64_t v) {
    q3 = q2; q2 = q1; q1 = q0; q0 = v;
    ++rem;
  }

  // words available without generating a block
  BK_HD int avail() const { return (int)rem + 4 * (int)cnt; }

  // Two Philox blocks at once (counters x and y), rounds interleaved: a block is a serial chain
  // of 10 rounds of 64x64->128 multiplies, and with one wavefront per SIMD nothing else hides
  // that latency -- two independent chains in flight do.
  BK_HD void block_at2(uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3, uint64_t y0, uint64_t y1,
                       uint64_t y2, uint64_t y3, uint64_t (&o)[4], uint64_t (&p)[4]) const {
    const uint64_t M0 = 0xD2E7470EE14C6C93ULL, M1 = 0xCA5A826395121157ULL;
    const uint64_t W0 = 0x9E3779B97F4A7C15ULL, W1 = 0xBB67AE8584CAA73BULL;
    uint64_t k0 = key0, k1 = key1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      if (r) { k0 += W0; k1 += W1; }
      uint64_t xh0, xl0, yh0, yl0, xh1, xl1, yh1, yl1;
      mulhilo64(M0, x0, xh0, xl0);
      mulhilo64(M0, y0, yh0, yl0);
      mulhilo64(M1, x2, xh1, xl1);
      mulhilo64(M1, y2, yh1, yl1);
      uint64_t xa = xor3_64(xh1, x1, k0), xc = xor3_64(xh0, x3, k1);
      uint64_t ya = xor3_64(yh1, y1, k0), yc = xor3_64(yh0, y3, k1);
      x0 = xa; x1 = xl1; x2 = xc; x3 = xl0;
      y0 = ya; y1 = yl1; y2 = yc; y3 = yl0;
    }
    o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
    p[0] = y0; p[1] = y1; p[2] = y2; p[3] = y3;
  }

  BK_HD static void ctr_add(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint64_t k, uint64_t& a0,
                            uint64_t& a1, uint64_t& a2, uint64_t& a3) {
    a0 = c0 + k;
    uint64_t cy = a0 < k ? 1ULL : 0ULL;
    a1 = c1 + cy;
    cy = (cy && a1 == 0) ? 1ULL : 0ULL;
    a2 = c2 + cy;
    cy = (cy && a2 == 0) ? 1ULL : 0ULL;
    a3 = c3 + cy;
  }

  // generate the next TWO lookahead blocks (caller guarantees cnt <= 1)
  BK_HD void prefetch() {
    uint64_t a0, a1, a2, a3, e0, e1, e2, e3;
    ctr_add(c0, c1, c2, c3, (uint64_t)cnt + 1, a0, a1, a2, a3);
    ctr_add(c0, c1, c2, c3, (uint64_t)cnt + 2, e0, e1, e2, e3);
    uint64_t t[4], u[4];
    block_at2(a0, a1, a2, a3, e0, e1, e2, e3, t, u);
    const bool first = (cnt == 0);  // cnt == 0: fill n1, n2 ; cnt == 1: fill n2, n3
    n10 = first ? t[0] : n10; n11 = first ? t[1] : n11; n12 = first ? t[2] : n12; n13 = first ? t[3] : n13;
    n20 = first ? u[0] : t[0]; n21 = first ? u[1] : t[1]; n22 = first ? u[2] : t[2]; n23 = first ? u[3] : t[3];
    n30 = first ? n30 : u[0]; n31 = first ? n31 : u[1]; n32 = first ? n32 : u[2]; n33 = first ? n33 : u[3];
    cnt += 2;
  }

  template <typename I>
  BK_HD void load(const uint64_t* st, I ld, I c) {
    key0 = st[0 * ld + c]; key1 = st[1 * ld + c];
    c0 = st[2 * ld + c]; c1 = st[3 * ld + c]; c2 = st[4 * ld + c]; c3 = st[5 * ld + c];
    b0 = st[6 * ld + c]; b1 = st[7 * ld + c]; b2 = st[8 * ld + c]; b3 = st[9 * ld + c];
    uint32_t pos = (uint32_t)st[10 * ld + c];
    rem = pos >= 4 ? 0u : 4u - pos;
    q0 = b0; q1 = b1; q2 = b2; q3 = b3;
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k)
      if (pos > k) { q0 = q1; q1 = q2; q2 = q3; }
    cnt = 0;
    n10 = n11 = n12 = n13 = n20 = n21 = n22 = n23 = n30 = n31 = n32 = n33 = 0;
  }
  template <typename I>
  BK_HD void store(uint64_t* st, I ld, I c) const {
    st[2 * ld + c] = c0; st[3 * ld + c] = c1; st[4 * ld + c] = c2; st[5 * ld + c] = c3;
    st[6 * ld + c] = b0; st[7 * ld + c] = b1; st[8 * ld + c] = b2; st[9 * ld + c] = b3;
    st[10 * ld + c] = 4u - rem;
  }
};

// ---- PCG64 (XSL-RR 128/64, numpy's default_rng(int)) --------------------------------
struct Pcg64 {
  uint64_t s_hi, s_lo, i_hi, i_lo;

  BK_HD uint64_t next() {
    const uint64_t MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
    // state = state * MULT + inc  (mod 2^128), then output from the NEW state
    uint64_t lo = s_lo * ML;
    uint64_t hi = mulhi64(s_lo, ML) + s_lo * MH + s_hi * ML;
    uint64_t nlo = lo + i_lo;
    uint64_t nhi = hi + i_hi + (nlo < lo ? 1ULL : 0ULL);
    s_lo = nlo; s_hi = nhi;
    uint64_t x = nhi ^ nlo;
    uint32_t rot = (uint32_t)(nhi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  BK_HD int avail() const { return 1 << 20; }  // one multiply per output: nothing to prefetch
  BK_HD void prefetch() {}
  static constexpr uint32_t cnt = 3;
  template <typename I>
  BK_HD void load(const uint64_t* st, I ld, I c) {
    s_hi = st[0 * ld + c]; s_lo = st[1 * ld + c]; i_hi = st[2 * ld + c]; i_lo = st[3 * ld + c];
  }
  template <typename I>
  BK_HD void store(uint64_t* st, I ld, I c) const {
    st[0 * ld + c] = s_hi; st[1 * ld + c] = s_lo;
  }
};

// ---- log1p for finite x > -1: Sun fdlibm 5.3 `s_log1p.c` ---------------------------------
// The algorithm, its thresholds and its constants are fdlibm's (the text FreeBSD msun and every libm since descend
// from): argument reduction 1 + x = 2^k (1 + f) with the correction term c, then log(1 + f) = f - f^2/2 + s (f^2/2 + R(z)),
// s = f / (2 + f), z = s^2, R a degree-7 polynomial in z with fdlibm's coefficients Lp1..Lp7.  Every operation is
// individually rounded (the library is built with -ffp-contract=off).  The ONE choice fdlibm leaves open that shows
// in the last bit is the order in which R's terms are summed: the reference's NumPy evaluates the ziggurat tail with
// the HOST libm's log1p, and the libm of this image sums R in pairs -- (Lp2 + z Lp3), (Lp4 + z Lp5), (Lp6 + z Lp7),
// weighted by z^2, z^4, z^6 -- rather than by Horner's rule; that order is used here so that the device returns the
// host's double, and tests/test_abi.py checks the equality on 20,000 arguments instead of assuming it.
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this software is freely granted,
//   provided that this notice is preserved.                       (see THIRD_PARTY.md)
BK_HD double bk_log1p(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01,
               Lp3 = 2.857142874366239149e-01, Lp4 = 2.222219843214978396e-01,
               Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  int32_t hx = (int32_t)(double_as_u64(x) >> 32);
  int32_t ax = hx & 0x7fffffff;
  int32_t k = 1, hu = 0;
  double f = 0.0, c = 0.0;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) return x == -1.0 ? -INFINITY : NAN;
    if (ax < 0x3e200000) {
      if (ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= (int32_t)0xbfd2bec3) { k = 0; f = x; hu = 1; }
  }
  if (k != 0) {
    double u = 1.0 + x;
    uint64_t ub = double_as_u64(u);
    hu = (int32_t)(ub >> 32);
    k = (hu >> 20) - 1023;
    c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
    c /= u;
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      ub = (ub & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32);
    } else {
      k += 1;
      ub = (ub & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32);
      hu = (0x00100000 - hu) >> 2;
    }
    f = u64_as_double(ub) - 1.0;
  }
  double hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    double R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  double s = f / (2.0 + f);
  double z = s * s;
  double R1 = z * Lp1, z2 = z * z;
  double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
  double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
  double R4 = Lp6 + z * Lp7;
  double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// Wave-synchronous top-up of the lookahead (device only; call where the whole wavefront
// is converged).  When any lane could run dry within one normal draw, every lane that has
// room generates one block at the same time.
template <typename G>
__device__ __forceinline__ void top_up(G& g) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__any(g.avail() < 4)) {
    if (g.cnt < 2) g.prefetch();  // two blocks, interleaved
  }
#endif
}

// ---- doubles and normals -------------------------------------------------------------
template <typename G>
BK_HD double next_double(G& g) {
  return (double)(g.next() >> 11) * (1.0 / 9007199254740992.0);
}

// NumPy random_standard_normal.  ki/wi/fi point at the three 256-entry tables (LDS on the
// device, static arrays on the host).  Split into the word-local fast test and the rare
// continuation so that two normals can be started from two words at once (next_normal_pair).
BK_HD bool zig_fast(uint64_t r, const uint64_t* ki, const double* wi, double& x, int& idx, uint64_t& rabs) {
  idx = (int)(r & 0xff);
  r >>= 8;
  int sign = (int)(r & 1);
  rabs = (r >> 1) & 0x000fffffffffffffULL;
  x = (double)rabs * wi[idx];
  if (sign) x = -x;
  return rabs < ki[idx];  // 99.2 %
}

// the rest of one ziggurat attempt whose first word failed the fast test; false = rejected
template <typename G>
BK_HD bool zig_slow(G& g, int idx, uint64_t rabs, double x, const double* fi, double& out) {
  const double zr = 3.6541528853610087963519472518, zinv = 0.27366123732975827203338247596;
  if (idx == 0) {
    for (;;) {
      double xx = -zinv * bk_log1p(-next_double(g));
      double yy = -bk_log1p(-next_double(g));
      if (yy + yy > xx * xx) {
        out = ((rabs >> 8) & 1) ? -(zr + xx) : zr + xx;
        return true;
      }
    }
  }
  if ((fi[idx - 1] - fi[idx]) * next_double(g) + fi[idx] < exp(-0.5 * x * x)) {
    out = x;
    return true;
  }
  return false;
}

template <typename G>
BK_HD double next_normal(G& g, const uint64_t* ki, const double* wi, const double* fi) {
  for (;;) {
    double x, out;
    int idx;
    uint64_t rabs;
    if (zig_fast(g.next(), ki, wi, x, idx, rabs)) return x;
    if (zig_slow(g, idx, rabs, x, fi, out)) return out;
  }
}

// Two consecutive normals.  Generic form: one after the other.
template <typename G>
BK_HD void next_normal_pair(G& g, const uint64_t* ki, const double* wi, const double* fi, double& z0, double& z1) {
  z0 = next_normal(g, ki, wi, fi);
  z1 = next_normal(g, ki, wi, fi);
}

// Philox: 98.3 % of the time both normals take one word each and pass the fast test, so the
// two words are popped and tested together (two independent table look-ups / conversions in
// flight instead of one -- the loop is latency bound with one wavefront per SIMD).  Otherwise
// the stream order is restored exactly: if the FIRST normal needs more words, the second word
// is pushed back and consumed again.
BK_HD void next_normal_pair(Philox& g, const uint64_t* ki, const double* wi, const double* fi, double& z0,
                            double& z1) {
  uint64_t r0 = g.next(), r1 = g.next();
  double x0, x1, out;
  int i0, i1;
  uint64_t a0, a1;
  bool ok0 = zig_fast(r0, ki, wi, x0, i0, a0);
  bool ok1 = zig_fast(r1, ki, wi, x1, i1, a1);
  if (ok0 && ok1) {
    z0 = x0;
    z1 = x1;
    return;
  }
  if (ok0) {
    z0 = x0;
    z1 = zig_slow(g, i1, a1, x1, fi, out) ? out : next_normal(g, ki, wi, fi);
    return;
  }
  g.unget(r1);
  z0 = zig_slow(g, i0, a0, x0, fi, out) ? out : next_normal(g, ki, wi, fi);
  z1 = next_normal(g, ki, wi, fi);
}

}  // namespace bk
