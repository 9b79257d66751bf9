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
 * The sequence (every line one IEEE-754 operation, rounded to nearest even; fma = the fused multiply-add of IEEE 754-2008,
 * ONE rounding -- v_fma_f64 on the device, fma() of the C library on the host, exact rational arithmetic in the oracle):
 *
 *     k = rint(x * log2(e))                                   x = k ln2 + r,  |r| <= ln2 / 2
 *     r = fma(-k, LN2_HI, x);  r = fma(-k, LN2_LO, r)         ln2 = LN2_HI + LN2_LO to 107 bits (the first fma is exact)
 *     q = 1/13!;  q = fma(q, r, 1/n!)  for n = 12 .. 2        Taylor: exp(r) = 1 + r + r^2 q, truncation < 0.04 ulp
 *     p = fma(r * r, q, r);  p = p + 1
 *     exp(x) = ldexp(p, k)                                     correctly rounded scaling, gradual underflow
 *
 * run on min(max(x, -746), 710): beyond those ldexp returns 0 / +inf by itself (so exp(x) = +inf from x = 709.7827128933841
 * on, 0 below -745.1332191019412); NaN gives NaN.  Measured against the host libm's (correctly rounded in all but a
 * handful of cases) exp: never more than 1 ulp off, equal for 91 % of arguments (tests/test_oracle_rng.py).
 * A sequence with a division (the classical rational approximation of exp) costs the funnel's one-wavefront-per-SIMD
 * trajectory kernels twice this one's time per leapfrog step (profiles/r6_cfg4_exp.md): on the device this is 20 instructions
 * without a branch.
 */
#ifndef BKHIP_MATH_H
#define BKHIP_MATH_H
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define BKHIP_MATH_FN static inline __host__ __device__
#else
#define BKHIP_MATH_FN static inline
#endif

/* One step of the polynomial: q r + c with ONE rounding.  On the device the three-operand form is written out: the compiler
 * otherwise picks the two-operand v_fmac_f64 and copies the coefficient into its destination first (11 extra moves per exp). */
#if defined(__HIP_DEVICE_COMPILE__)
static inline __device__ double bk_fma_(double a, double b, double c) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}
#else
static inline double bk_fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif

BKHIP_MATH_FN double bk_exp(double x) {
  const double log2e = 0x1.71547652b82fep+0, ln2_hi = 0x1.62e42fefa39efp-1, ln2_lo = 0x1.abc9e3b39803fp-56;
  /* 1/n!, n = 2 .. 13, each rounded to nearest */
  const double c2 = 0x1.0000000000000p-1, c3 = 0x1.5555555555555p-3, c4 = 0x1.5555555555555p-5,
               c5 = 0x1.1111111111111p-7, c6 = 0x1.6c16c16c16c17p-10, c7 = 0x1.a01a01a01a01ap-13,
               c8 = 0x1.a01a01a01a01ap-16, c9 = 0x1.71de3a556c734p-19, c10 = 0x1.27e4fb7789f5cp-22,
               c11 = 0x1.ae64567f544e4p-26, c12 = 0x1.1eed8eff8d898p-29, c13 = 0x1.6124613a86d09p-33;
  /* [-746, 710] holds every argument whose exp is neither 0 nor +inf; at the ends ldexp gives exactly those (NaN: below) */
  const double xs = __builtin_fmin(__builtin_fmax(x, -746.0), 710.0);
  const double k = __builtin_rint(xs * log2e);
  double r = __builtin_fma(-k, ln2_hi, xs);
  r = __builtin_fma(-k, ln2_lo, r);
  double q = c13;
  q = bk_fma_(q, r, c12);
  q = bk_fma_(q, r, c11);
  q = bk_fma_(q, r, c10);
  q = bk_fma_(q, r, c9);
  q = bk_fma_(q, r, c8);
  q = bk_fma_(q, r, c7);
  q = bk_fma_(q, r, c6);
  q = bk_fma_(q, r, c5);
  q = bk_fma_(q, r, c4);
  q = bk_fma_(q, r, c3);
  q = bk_fma_(q, r, c2);
  double p = __builtin_fma(r * r, q, r);
  p = p + 1.0;
  const double y = __builtin_ldexp(p, (int)k);
  return x != x ? x + x : y;
}

#endif /* BKHIP_MATH_H */
