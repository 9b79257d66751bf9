// Built-in target densities: the batched, device-resident counterpart of
// GradModel.log_density_gradient (bayes_kit/typing.py:25-27) for targets whose gradient the
// library can evaluate itself ("thin C-ABI callback").  Operation order follows
// oracle/models.py exactly (no FMA contraction).
//
// Inside a trajectory the log density is not needed (hmc.py:45,50 discard it), so the
// gradient-only form of the separable Gaussians is a pure streaming elementwise kernel
// (16 algorithmic bytes per element: read theta, write grad).  When logp is requested the
// per-chain sum runs sequentially over d in one lane.
#include "bk_common.hpp"

namespace {

constexpr int TG_ROWS = 4;
constexpr int TG_BLOCK = 256;
constexpr int PC_BLOCK = 64;
constexpr int PC_UNROLL = 8;

// grad = -(lam*theta)  (lam NULL -> grad = -theta), two chains per lane
typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int ROWS, bool NT>
__global__ __launch_bounds__(TG_BLOCK) void k_gauss_grad_v2(const double* th, double* g, i64 ld,
                                                            const double* lam, i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * TG_BLOCK + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  dvec2 t[ROWS];
  double l[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
    if (d0 + i < D) {
      const dvec2* p = reinterpret_cast<const dvec2*>(th + (d0 + i) * ld + 2 * c2);
      t[i] = NT ? __builtin_nontemporal_load(p) : *p;
      l[i] = lam ? lam[d0 + i] : 1.0;
    }
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
    if (d0 + i < D) {
      dvec2 o;
      o.x = lam ? -(l[i] * t[i].x) : -t[i].x;
      o.y = lam ? -(l[i] * t[i].y) : -t[i].y;
      dvec2* q = reinterpret_cast<dvec2*>(g + (d0 + i) * ld + 2 * c2);
      if (NT) __builtin_nontemporal_store(o, q);
      else *q = o;
    }
}

__global__ __launch_bounds__(TG_BLOCK) void k_gauss_grad_s(const double* th, double* g, i64 ld,
                                                           const double* lam, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * TG_BLOCK + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * TG_ROWS;
  if (c >= C) return;
#pragma unroll
  for (int i = 0; i < TG_ROWS; ++i)
    if (d0 + i < D) {
      double t = th[(d0 + i) * ld + c];
      g[(d0 + i) * ld + c] = lam ? -(lam[d0 + i] * t) : -t;
    }
}

// logp (and optionally grad) of the separable Gaussians, one lane per chain
__global__ __launch_bounds__(PC_BLOCK) void k_gauss_logp(const double* th, double* g, double* logp, i64 ld,
                                                         const double* lam, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (i64 d0 = 0; d0 < D; d0 += PC_UNROLL) {
    double t[PC_UNROLL];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) t[u] = th[(d0 + u) * ld + c];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) {
        double lt = lam ? lam[d0 + u] * t[u] : t[u];
        s = s + t[u] * lt;
        if (g) g[(d0 + u) * ld + c] = -lt;
      }
  }
  logp[c] = -0.5 * s;
}

// Neal's funnel
__global__ __launch_bounds__(PC_BLOCK) void k_funnel(const double* th, double* g, double* logp, i64 ld,
                                                     i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  double v = th[c];
  double s = 0.0;
  for (i64 d0 = 1; d0 < D; d0 += PC_UNROLL) {
    double t[PC_UNROLL];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) t[u] = th[(d0 + u) * ld + c];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) s = s + t[u] * t[u];
  }
  double ev = exp(-v);
  double hn = 0.5 * (double)(D - 1);
  double he = 0.5 * ev;
  if (logp) logp[c] = ((-(v * v) / 18.0) - hn * v) - he * s;
  if (g) {
    g[c] = ((-v / 9.0) - hn) + he * s;
    for (i64 d0 = 1; d0 < D; d0 += PC_UNROLL) {
      double t[PC_UNROLL];
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < D) t[u] = th[(d0 + u) * ld + c];
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < D) g[(d0 + u) * ld + c] = -(ev * t[u]);
    }
  }
}

int gauss(const double* theta, double* grad, double* logp, i64 ld, const double* lam, i64 C, i64 D,
          void* stream) {
  if (!theta || (!grad && !logp) || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (logp) {
    k_gauss_logp<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, s>>>(theta, grad, logp, ld, lam, C, D);
    BK_RETURN_LAUNCH_STATUS();
  }
  if (D == 0) return BK_OK;
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(theta) && bk_aligned16(grad)) {
    if (bk_streams_past_llc(2 * C * D)) {
      dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)bk_cdiv(D, 4));
      k_gauss_grad_v2<4, true><<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C / 2, D);
    } else {
      dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)bk_cdiv(D, 2));
      k_gauss_grad_v2<2, false><<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C / 2, D);
    }
  } else {
    dim3 grid((unsigned)bk_cdiv(C, TG_BLOCK), (unsigned)bk_cdiv(D, TG_ROWS));
    k_gauss_grad_s<<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

}  // namespace

extern "C" {

int bk_target_iso_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                                int64_t D, void* stream) {
  return gauss(theta, grad, logp, ld, nullptr, C, D, stream);
}

int bk_target_diag_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld,
                                 const double* lam, int64_t C, int64_t D, void* stream) {
  if (!lam) return BK_E_ARG;
  return gauss(theta, grad, logp, ld, lam, C, D, stream);
}

int bk_target_funnel_grad(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                          int64_t D, void* stream) {
  if (!theta || (!grad && !logp) || C < 0 || D < 1) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_funnel<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(theta, grad, logp,
                                                                                         ld, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
