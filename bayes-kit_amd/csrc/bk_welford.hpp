// Welford update of the per-chain running moments (bayes_kit/rhat.py:111-171 consumes them), shared by the kernels of
// bk_diag.hip and by the generator launch that carries the update along as a side job (bk_rng.hip: k_zig_parallel_side).
#pragma once
#include "bk_common.hpp"

namespace bkw {

constexpr int EL_ROWS = 4;

// Welford: after the n-th draw, mean = np.mean(draws[:n]) and m2/(n-1) = np.var(ddof=1)
__device__ __forceinline__ void welford_elem(double x, double& mu, double& q, double n) {
  double delta = x - mu;
  mu = mu + delta / n;
  q = q + delta * (x - mu);
}

typedef double dvec2 __attribute__((ext_vector_type(2)));

// one unit = 256 lanes x 2 chains x EL_ROWS dimensions: two chains (16 B) per lane, 40 algorithmic bytes per element
// (R theta, mean, M2; W mean, M2); non-temporal when the three arrays stream past the Infinity Cache.
// (ld_th: theta's own row pitch; bx / by: the unit's column block / row block; tid: lane of the unit, 0..255)
template <bool NT>
__device__ __forceinline__ void welford_unit_v2(i64 bx, i64 by, int tid, double* mean, double* m2, const double* th, i64 ld,
                                                i64 ld_th, double n, i64 C2, i64 D) {
  i64 c2 = bx * 256 + tid;
  i64 d0 = by * EL_ROWS;
  if (c2 >= C2) return;
  dvec2 x[EL_ROWS], mu[EL_ROWS], q[EL_ROWS];
#pragma unroll
  for (int i = 0; i < EL_ROWS; ++i)
    if (d0 + i < D) {
      i64 o = (d0 + i) * ld + 2 * c2;
      const dvec2 *px = reinterpret_cast<const dvec2*>(th + (d0 + i) * ld_th + 2 * c2), *pm = reinterpret_cast<const dvec2*>(mean + o),
                  *pq = reinterpret_cast<const dvec2*>(m2 + o);
      x[i] = NT ? __builtin_nontemporal_load(px) : *px;
      mu[i] = NT ? __builtin_nontemporal_load(pm) : *pm;
      q[i] = NT ? __builtin_nontemporal_load(pq) : *pq;
    }
#pragma unroll
  for (int i = 0; i < EL_ROWS; ++i)
    if (d0 + i < D) {
      double m0 = mu[i].x, m1 = mu[i].y, q0 = q[i].x, q1 = q[i].y;
      welford_elem(x[i].x, m0, q0, n);
      welford_elem(x[i].y, m1, q1, n);
      mu[i] = dvec2{m0, m1};
      q[i] = dvec2{q0, q1};
      i64 o = (d0 + i) * ld + 2 * c2;
      dvec2 *pm = reinterpret_cast<dvec2*>(mean + o), *pq = reinterpret_cast<dvec2*>(m2 + o);
      if (NT) {
        __builtin_nontemporal_store(mu[i], pm);
        __builtin_nontemporal_store(q[i], pq);
      } else {
        *pm = mu[i];
        *pq = q[i];
      }
    }
}

// whether the two-chains-per-lane unit applies (welford_launch's test)
static inline bool v2_applies(const double* mean, const double* m2, const double* theta, i64 ld, i64 ld_th, i64 C) {
  return C % 2 == 0 && ld % 2 == 0 && ld_th % 2 == 0 && bk_aligned16(mean) && bk_aligned16(m2) && bk_aligned16(theta);
}

}  // namespace bkw
