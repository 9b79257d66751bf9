/* bkhip_math.h -- the library's own exp(), the same double on the host and on the device.
 *
 * Why: densities with an exp() inside a gradient that feeds a chaotic flow (Neal's funnel, bayes_kit/drghmc.py:253-289 on
 * it) amplify a one-ulp difference between two correctly-working exp() implementations by 10-100 per trajectory; the
 * device's math library and the host's disagree in the last bit for ~8 % of arguments.  With bk_exp() the library's funnel
 * (and any CTarget / from-source density that calls it) evaluates a SPECIFIED sequence of individually rounded IEEE
 * operations, which oracle/rng.py restates operation for operation: the HIP kernels and the NumPy oracle then agree bit for
 * bit on the funnel fixtures (tests/test_gpu_samplers.py), and what is left between them and the reference is the
 * reference's own libm / BLAS.
 *
 * The algorithm, its thresholds and its constants are Sun fdlibm 5.3 `e_exp.c` (__ieee754_exp): argument reduction
 * x = k ln2 + r with ln2 split in two, r = hi - lo, the rational approximation exp(r) = 1 + r + r c / (2 - c) with
 * c = r - r^2 (P1 + r^2 (P2 + ... P5)), scaling by 2^k through the exponent field.  < 1 ulp.  Compile with
 * -ffp-contract=off (as the library is): every * and + below is one rounding.
 *   Copyright (C) 2004 by Sun Microsystems, Inc. All rights reserved.
 *   Permission to use, copy, modify, and distribute this software is freely granted,
 *   provided that this notice is preserved.                                   (see THIRD_PARTY.md)
 */
#ifndef BKHIP_MATH_H
#define BKHIP_MATH_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define BKHIP_MATH_FN static inline __host__ __device__
#else
#define BKHIP_MATH_FN static inline
#endif

BKHIP_MATH_FN double bk_exp(double x) {
  const double one = 1.0, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302, /* 2**-1000 */
      o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
      ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
      P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  uint32_t hx = (uint32_t)(bits >> 32);
  const uint32_t lx = (uint32_t)bits;
  const int xsb = (int)((hx >> 31) & 1u); /* sign bit of x */
  hx &= 0x7fffffffu;                      /* high word of |x| */
  double hi = 0.0, lo = 0.0, t, c, y;
  int k = 0;
  if (hx >= 0x40862E42u) { /* |x| >= 709.78... */
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | lx) != 0) return x + x; /* NaN */
      return xsb == 0 ? x : 0.0;                     /* exp(+-inf) = {inf, 0} */
    }
    if (x > o_threshold) return huge * huge;         /* overflow */
    if (x < u_threshold) return twom1000 * twom1000; /* underflow */
  }
  if (hx > 0x3fd62e42u) {   /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2u) { /* and |x| < 1.5 ln2 */
      hi = xsb ? x + ln2HI : x - ln2HI;
      lo = xsb ? -ln2LO : ln2LO;
      k = 1 - xsb - xsb;
    } else {
      k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
      t = (double)k;
      hi = x - t * ln2HI; /* t*ln2HI is exact here */
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) { /* |x| < 2**-28 */
    if (huge + x > one) return one + x;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return one - ((x * c) / (c - 2.0) - x);
  y = one - ((lo - (x * c) / (2.0 - c)) - hi);
  memcpy(&bits, &y, 8);
  if (k >= -1021) {
    bits += (uint64_t)(int64_t)k << 52; /* add k to y's exponent */
    memcpy(&y, &bits, 8);
    return y;
  }
  bits += (uint64_t)(int64_t)(k + 1000) << 52;
  memcpy(&y, &bits, 8);
  return y * twom1000;
}

#endif /* BKHIP_MATH_H */
