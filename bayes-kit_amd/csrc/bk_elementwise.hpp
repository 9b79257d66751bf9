// Separable densities (log p = sum over coordinates of a term of that coordinate alone): the library's whole-trajectory
// and whole-draw HMC kernels for ANY such density.  The density is the only model-specific code: a struct with
//
//     static void   eval(double th, i64 d, const double* params, double& term, double& grad);
//     static double finish(double s);      // log p from the sum of the terms (identity, or a scale such as -1/2)
//
// Every (d, c) element is independent of all others through the L steps of hmc.py:40-53, so theta and the momentum are
// read once and written once per TRAJECTORY and the kernels are bound by the fp64 vector units, not by HBM.  The
// per-element operation sequence is exactly the one the step-by-step kernels execute (bk_leapfrog_kick_drift around the
// target's gradient op), so the results are bit-identical to that path.
// Instantiated for CTarget.from_source(form="elementwise") densities (the generated translation unit includes this file;
// the user's bk_term is the only inlined callee).
#pragma once
#include "bk_common.hpp"

namespace bke {

constexpr int BLOCK = 256;
typedef double dvec2 __attribute__((ext_vector_type(2)));

// Whole trajectory (hmc.py:40-53), theta and rho out: two chains (16 B) per lane, ROWS rows per thread advancing
// together (2*ROWS independent dependency chains per lane).
template <class TERM, int ROWS, bool HM>
__global__ __launch_bounds__(BLOCK) void k_traj(const double* th_in, double* th_out, const double* rho_in, double* rho_out,
                                                i64 ld, const double* params, const double* metric, double eps, int steps,
                                                i64 C2, i64 D) {
  const i64 c2 = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  const double half = 0.5 * eps;
  dvec2 th[ROWS], r[ROWS], t[ROWS];
  double m[ROWS];
  i64 dd[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    dd[i] = (d0 + i < D) ? d0 + i : D - 1;  // rows past the end recompute the last one, not stored
    th[i] = *reinterpret_cast<const dvec2*>(th_in + dd[i] * ld + 2 * c2);
    r[i] = *reinterpret_cast<const dvec2*>(rho_in + dd[i] * ld + 2 * c2);
    m[i] = HM ? metric[dd[i]] : 1.0;
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    double term, gx, gy;
    TERM::eval(th[i].x, dd[i], params, term, gx);
    TERM::eval(th[i].y, dd[i], params, term, gy);
    t[i].x = HM ? m[i] * gx : gx;
    t[i].y = HM ? m[i] * gy : gy;
    r[i].x = r[i].x + (-half) * t[i].x;  // hmc.py:46
    r[i].y = r[i].y + (-half) * t[i].y;
  }
  for (int n = 0; n < steps; ++n) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      r[i].x = r[i].x + eps * t[i].x;  // hmc.py:48
      r[i].y = r[i].y + eps * t[i].y;
      th[i].x = th[i].x + eps * r[i].x;  // hmc.py:49
      th[i].y = th[i].y + eps * r[i].y;
      double term, gx, gy;  // hmc.py:50
      TERM::eval(th[i].x, dd[i], params, term, gx);
      TERM::eval(th[i].y, dd[i], params, term, gy);
      t[i].x = HM ? m[i] * gx : gx;
      t[i].y = HM ? m[i] * gy : gy;
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    r[i].x = r[i].x + half * t[i].x;  // hmc.py:52
    r[i].y = r[i].y + half * t[i].y;
    if (d0 + i < D) {
      *reinterpret_cast<dvec2*>(th_out + (d0 + i) * ld + 2 * c2) = th[i];
      *reinterpret_cast<dvec2*>(rho_out + (d0 + i) * ld + 2 * c2) = r[i];
    }
  }
}

template <class TERM>
__global__ __launch_bounds__(BLOCK) void k_traj_s(const double* th_in, double* th_out, const double* rho_in, double* rho_out,
                                                  i64 ld, const double* params, const double* metric, double eps, int steps,
                                                  i64 C, i64 D) {
  const i64 c = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 d = blockIdx.y;
  if (c >= C) return;
  const double half = 0.5 * eps;
  double th = th_in[d * ld + c], r = rho_in[d * ld + c];
  const double m = metric ? metric[d] : 1.0;
  double term, g;
  TERM::eval(th, d, params, term, g);
  double t = metric ? m * g : g;
  r = r + (-half) * t;
  for (int n = 0; n < steps; ++n) {
    r = r + eps * t;
    th = th + eps * r;
    TERM::eval(th, d, params, term, g);
    t = metric ? m * g : g;
  }
  r = r + half * t;
  th_out[d * ld + c] = th;
  rho_out[d * ld + c] = r;
}

// The same trajectory as the trajectory + energies of a whole HMC draw (hmc.py:55-59): one thread per (chain, QUARTER of
// the dimensions) walks its quarter in chunks of TQ_ROWS rows -- each chunk a register-resident trajectory as above --
// and accumulates, sequentially in d, the three per-chain sums a draw needs:
//     kin0 = 1/2 sum rho0*(m*rho0)   [hmc.py:57 -> :37]    (momentum as drawn)
//     kin1 = 1/2 sum rho1*(m*rho1)   [hmc.py:59 -> :37]    (momentum at the end)
//     lp1  = finish(sum of the terms at theta1)            (the target's log density at the end)
// as quarter partials part[k][q][c]; combined ((p0+p1)+p2)+p3 (k_quarter_sums) they are, bit for bit, what
// bk_leapfrog_finish and the target's log-density op compute with their four wavefronts per 64 chains -- which is why the
// walk is per quarter and in d order.  The momentum is read either in the state layout (rho_in) or straight from the
// wavefront-per-chain generator's chain-major normals (zt[c*ldz + d], rho0 = 0.0 + 1.0*z as numpy's random_normal); the
// final momentum is never stored (HMC discards it).
constexpr int TQ_ROWS = 8;
template <class TERM, bool HM, bool ZT>
__global__ __launch_bounds__(BLOCK) void k_traj_q(const double* th_in, double* th_out, const double* rho_in, i64 ld,
                                                  const double* zt, i64 ldz, const double* params, const double* metric,
                                                  double eps, int steps, double* part, i64 C, i64 D) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;  // (64-thread workgroups for small launches)
  const int q = blockIdx.y;
  if (c >= C) return;
  const i64 Dq = (D + 3) / 4;
  const i64 dlo = q * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  const double half = 0.5 * eps;
  double k0 = 0.0, k1 = 0.0, sl = 0.0;
  for (i64 d0 = dlo; d0 < dhi; d0 += TQ_ROWS) {
    double th[TQ_ROWS], r[TQ_ROWS], t[TQ_ROWS], m[TQ_ROWS];
    i64 dd[TQ_ROWS];
#pragma unroll
    for (int i = 0; i < TQ_ROWS; ++i) {
      const i64 d = (d0 + i < dhi) ? d0 + i : dhi - 1;  // rows past the end recompute the last one, unused
      dd[i] = d;
      th[i] = th_in[d * ld + c];
      r[i] = ZT ? 0.0 + 1.0 * zt[c * ldz + d] : rho_in[d * ld + c];
      m[i] = HM ? metric[d] : 1.0;
    }
#pragma unroll
    for (int i = 0; i < TQ_ROWS; ++i) {
      if (d0 + i < dhi) {
        const double mv = HM ? m[i] * r[i] : r[i];
        k0 = k0 + r[i] * mv;
      }
      double term, g;
      TERM::eval(th[i], dd[i], params, term, g);
      t[i] = HM ? m[i] * g : g;
      r[i] = r[i] + (-half) * t[i];  // hmc.py:46
    }
    for (int n = 0; n < steps; ++n) {
#pragma unroll
      for (int i = 0; i < TQ_ROWS; ++i) {
        r[i] = r[i] + eps * t[i];    // hmc.py:48
        th[i] = th[i] + eps * r[i];  // hmc.py:49
        double term, g;              // hmc.py:50
        TERM::eval(th[i], dd[i], params, term, g);
        t[i] = HM ? m[i] * g : g;
      }
    }
#pragma unroll
    for (int i = 0; i < TQ_ROWS; ++i) {
      r[i] = r[i] + half * t[i];  // hmc.py:52
      if (d0 + i < dhi) {
        th_out[(d0 + i) * ld + c] = th[i];
        const double mv = HM ? m[i] * r[i] : r[i];
        k1 = k1 + r[i] * mv;
        double term, g;
        TERM::eval(th[i], dd[i], params, term, g);
        sl = sl + term;
      }
    }
  }
  part[(0 * 4 + q) * C + c] = k0;
  part[(1 * 4 + q) * C + c] = k1;
  part[(2 * 4 + q) * C + c] = sl;
}

// ... and, when the caller hands over the rest of the accept test's inputs, the test itself
// (hmc.py:60-63, the arithmetic of bk_mh_accept in HMC mode): one launch less per draw.
template <class TERM>
__global__ __launch_bounds__(256) void k_quarter_sums(const double* part, double* kin0, double* kin1, double* lp,
                                                      double* lp_cur, const double* log_u, uint8_t* mask, double* ret,
                                                      uint32_t* count, i64 C) {
  const i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  bool acc = false;
  if (c < C) {
    double v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double* p = part + (i64)k * 4 * C + c;
      v[k] = ((p[0] + p[C]) + p[2 * C]) + p[3 * C];
    }
    const double a0 = 0.5 * v[0], a1 = 0.5 * v[1], l1 = TERM::finish(v[2]);
    if (kin0) kin0[c] = a0;
    kin1[c] = a1;
    lp[c] = l1;
    if (lp_cur) {
      const double l0 = lp_cur[c];
      const double h0 = l0 - a0, h1 = l1 - a1;  // hmc.py:36-38
      acc = log_u[c] < h1 - h0;                 // hmc.py:60
      if (mask) mask[c] = acc ? 1 : 0;
      if (ret) ret[c] = acc ? h1 : h0;
      if (acc) lp_cur[c] = l1;
    }
  }
  if (lp_cur && count) {
    unsigned long long b = __ballot(acc);
    if ((threadIdx.x & (BK_WAVE - 1)) == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
  }
}

// One leapfrog step {gradient, kick, drift} (hmc.py:48-50, drghmc.py:280-283) of a separable density as ONE streaming launch:
// rho += h * (metric * grad(theta)); theta += h * rho, in place.  Every element is independent: two chains (16 B) per lane,
// ROWS rows per thread; 32 D algorithmic bytes per chain-step (read and write theta and rho) instead of the 56 D of a gradient
// launch followed by bk_leapfrog_kick_drift, the same arithmetic (bit-identical), half the launches.
template <class TERM, int ROWS, bool HM, bool NT>
__global__ __launch_bounds__(BLOCK) void k_step(double* th, double* rho, i64 ld, const double* params, const double* metric,
                                                double h, i64 C2, i64 D) {
  const i64 c2 = (i64)blockIdx.x * BLOCK + threadIdx.x;
  const i64 d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  dvec2 t[ROWS], r[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
    if (d0 + i < D) {
      const dvec2* pt = reinterpret_cast<const dvec2*>(th + (d0 + i) * ld + 2 * c2);
      const dvec2* pr = reinterpret_cast<const dvec2*>(rho + (d0 + i) * ld + 2 * c2);
      t[i] = NT ? __builtin_nontemporal_load(pt) : *pt;
      r[i] = NT ? __builtin_nontemporal_load(pr) : *pr;
    }
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
    if (d0 + i < D) {
      double term, gx, gy;
      TERM::eval(t[i].x, d0 + i, params, term, gx);
      TERM::eval(t[i].y, d0 + i, params, term, gy);
      const double m = HM ? metric[d0 + i] : 1.0;
      const double tx = HM ? m * gx : gx, ty = HM ? m * gy : gy;
      r[i].x = r[i].x + h * tx;
      r[i].y = r[i].y + h * ty;
      t[i].x = t[i].x + h * r[i].x;
      t[i].y = t[i].y + h * r[i].y;
      dvec2* qt = reinterpret_cast<dvec2*>(th + (d0 + i) * ld + 2 * c2);
      dvec2* qr = reinterpret_cast<dvec2*>(rho + (d0 + i) * ld + 2 * c2);
      if (NT) {
        __builtin_nontemporal_store(r[i], qr);
        __builtin_nontemporal_store(t[i], qt);
      } else {
        *qr = r[i];
        *qt = t[i];
      }
    }
}

// one chain per lane: odd shapes, unaligned views, and every launch whose chain count lives on the device
template <class TERM>
__global__ __launch_bounds__(BLOCK) void k_step_s(double* th, double* rho, i64 ld, const double* params, const double* metric,
                                                  double h, i64 C_host, i64 D, const uint32_t* n_dev) {
  const i64 C = bk_lanes(C_host, n_dev);
  const i64 c = (i64)blockIdx.x * BLOCK + threadIdx.x, d0 = (i64)blockIdx.y * 4;
  if (c >= C) return;
  double t[4], r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) {
      t[i] = th[(d0 + i) * ld + c];
      r[i] = rho[(d0 + i) * ld + c];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) {
      double term, g;
      TERM::eval(t[i], d0 + i, params, term, g);
      const double tt = metric ? metric[d0 + i] * g : g;
      r[i] = r[i] + h * tt;
      rho[(d0 + i) * ld + c] = r[i];
      th[(d0 + i) * ld + c] = t[i] + h * r[i];
    }
}

// Host side: theta, rho [D][ld] advanced in place by one leapfrog step of size h over min(n, *n_dev) chains.
template <class TERM>
static int step_launch(double* theta, double* rho, int64_t ld, const double* params, const double* metric, double h, int64_t n,
                       int64_t D, const uint32_t* n_dev, void* stream) {
  if (!theta || !rho || n < 0 || D < 0) return BK_E_ARG;
  if (ld < n) return BK_E_ALIGN;
  if (n == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  const bool vec = !n_dev && n % 2 == 0 && ld % 2 == 0 && bk_aligned16(theta) && bk_aligned16(rho);
  if (vec) {
    const bool nt = bk_streams_past_llc(4 * n * D) && D <= 65535;
    if (nt) {  // (two rows per thread: 0.82 of 8 TB/s at config-3 shape; one: 0.79, four: 0.70 -- tools/step_stream_bench.py)
      dim3 grid((unsigned)bk_cdiv(n / 2, BLOCK), (unsigned)bk_cdiv(D, 2));
      if (metric) k_step<TERM, 2, true, true><<<grid, dim3(BLOCK), 0, s>>>(theta, rho, ld, params, metric, h, n / 2, D);
      else k_step<TERM, 2, false, true><<<grid, dim3(BLOCK), 0, s>>>(theta, rho, ld, params, metric, h, n / 2, D);
    } else {
      dim3 grid((unsigned)bk_cdiv(n / 2, BLOCK), (unsigned)bk_cdiv(D, 2));
      if (metric) k_step<TERM, 2, true, false><<<grid, dim3(BLOCK), 0, s>>>(theta, rho, ld, params, metric, h, n / 2, D);
      else k_step<TERM, 2, false, false><<<grid, dim3(BLOCK), 0, s>>>(theta, rho, ld, params, metric, h, n / 2, D);
    }
  } else {
    dim3 grid((unsigned)bk_cdiv(n, BLOCK), (unsigned)bk_cdiv(D, 4));
    k_step_s<TERM><<<grid, dim3(BLOCK), 0, s>>>(theta, rho, ld, params, metric, h, n, D, n_dev);
  }
  BK_RETURN_LAUNCH_STATUS();
}

// Host side of bk_hmc_trajectory_gaussian for any separable density (include/bkhip.h documents the arguments).
template <class TERM>
static int hmc_trajectory_launch(const double* theta_in, double* theta_out, const double* rho_in, double* rho_out, int64_t ld,
                                 const double* params, const double* metric, double eps, int64_t steps, int64_t C, int64_t D,
                                 void* stream) {
  if (!theta_in || !theta_out || !rho_in || !rho_out || steps < 0 || steps > 0x7fffffff || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(theta_in) && bk_aligned16(theta_out) && bk_aligned16(rho_in) &&
      bk_aligned16(rho_out)) {
    constexpr int ROWS = 4;
    dim3 grid((unsigned)bk_cdiv(C / 2, BLOCK), (unsigned)bk_cdiv(D, ROWS));
    if (metric)
      k_traj<TERM, ROWS, true><<<grid, dim3(BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, params, metric, eps,
                                                            (int)steps, C / 2, D);
    else
      k_traj<TERM, ROWS, false><<<grid, dim3(BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, params, metric, eps,
                                                             (int)steps, C / 2, D);
  } else {
    dim3 grid((unsigned)bk_cdiv(C, BLOCK), (unsigned)D);
    k_traj_s<TERM><<<grid, dim3(BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, params, metric, eps, (int)steps, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

// Host side of bk_hmc_draw_gaussian for any separable density.
template <class TERM>
static int hmc_draw_launch(const double* theta_in, double* theta_out, int64_t ld, const double* rho_in, const double* zt,
                           int64_t ldz, const double* params, const double* metric, double eps, int64_t steps, double* part,
                           double* kin0, double* kin1, double* lp_out, double* lp_cur, const double* log_u,
                           uint8_t* accept_mask, double* ret, uint32_t* accept_count, int64_t C, int64_t D, void* stream) {
  if (!theta_in || !theta_out || (!rho_in && !zt) || (rho_in && zt) || !part || !kin1 || !lp_out || steps < 0 ||
      steps > 0x7fffffff || C < 0 || D < 0 || (lp_cur && !log_u))
    return BK_E_ARG;
  if (ld < C || (zt && ldz < D)) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  // one wavefront per workgroup while that still leaves CUs idle (4096 chains: 256 workgroups instead of 64)
  const int tq_block = C * 4 <= 256 * BLOCK ? BK_WAVE : BLOCK;
  dim3 grid((unsigned)bk_cdiv(C, tq_block), 4);
#define BKE_TQ(HM, ZT)                                                                                              \
  k_traj_q<TERM, HM, ZT><<<grid, dim3(tq_block), 0, s>>>(theta_in, theta_out, rho_in, ld, zt, ldz, params, metric, eps, \
                                                         (int)steps, part, C, D)
  if (metric && zt) BKE_TQ(true, true);
  else if (metric) BKE_TQ(true, false);
  else if (zt) BKE_TQ(false, true);
  else BKE_TQ(false, false);
#undef BKE_TQ
  k_quarter_sums<TERM><<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, s>>>(part, kin0, kin1, lp_out, lp_cur, log_u,
                                                                             accept_mask, ret, accept_count, C);
  BK_RETURN_LAUNCH_STATUS();
}

}  // namespace bke
