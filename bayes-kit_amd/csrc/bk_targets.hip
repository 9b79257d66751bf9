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
#include "bk_elementwise.hpp"
#include "bk_mala_step.hpp"
#include "bk_lanes.hpp"
#include <stdlib.h>

namespace {

constexpr int TG_ROWS = 4;
constexpr int TG_BLOCK = 256;

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
                                                           const double* lam, i64 C_host, i64 D,
                                                           const uint32_t* n_dev) {
  const i64 C = bk_lanes(C_host, n_dev);
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

// logp (and optionally grad) of the separable Gaussians: 4 wavefronts per 64 chains, wavefront w
// sums its contiguous quarter of the dimensions sequentially, quarters combined in fixed order
constexpr int RED_WAVES = 4;
constexpr int RED_BLOCK = RED_WAVES * BK_WAVE;
__global__ __launch_bounds__(RED_BLOCK) void k_gauss_logp(const double* th, double* g, double* logp, i64 ld,
                                                          const double* lam, i64 C_host, i64 D,
                                                          const uint32_t* n_dev) {
  __shared__ double part[RED_WAVES][BK_WAVE];
  constexpr int PC_UNROLL = 8;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 C = bk_lanes(C_host, n_dev);
  if ((i64)blockIdx.x * BK_WAVE >= C) return;  // (whole workgroup past the set: uniform)
  const i64 c = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 Dq = (D + RED_WAVES - 1) / RED_WAVES;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  double s = 0.0;
  if (c < C) {
    for (i64 d0 = dlo; d0 < dhi; d0 += PC_UNROLL) {
      double t[PC_UNROLL];
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < dhi) t[u] = th[(d0 + u) * ld + c];
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < dhi) {
          double lt = lam ? lam[d0 + u] * t[u] : t[u];
          s = s + t[u] * lt;
          if (g) g[(d0 + u) * ld + c] = -lt;
        }
    }
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    double tot = part[0][lane];
#pragma unroll
    for (int k = 1; k < RED_WAVES; ++k) tot = tot + part[k][lane];
    logp[c] = -0.5 * tot;
  }
}

// ... two chains (16 B) per lane: 128 chains per workgroup, 1 KiB per wavefront and row (the 8-byte form
// streams at 5.2 TB/s, 207 us per config-3 launch).  Per component the scalar kernel's operation sequence.
__global__ __launch_bounds__(RED_BLOCK) void k_gauss_logp_v2(const double* th, double* g, double* logp, i64 ld,
                                                             const double* lam, i64 C2, i64 D) {
  __shared__ dvec2 part[RED_WAVES][BK_WAVE];
  constexpr int U = 8;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 c2 = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 Dq = (D + RED_WAVES - 1) / RED_WAVES;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  dvec2 s = {0.0, 0.0};
  if (c2 < C2) {
    for (i64 d0 = dlo; d0 < dhi; d0 += U) {
      dvec2 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) t[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(th + (d0 + u) * ld + 2 * c2));
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) {
          const double l = lam ? lam[d0 + u] : 1.0;
          const double lx = lam ? l * t[u].x : t[u].x, ly = lam ? l * t[u].y : t[u].y;
          s.x = s.x + t[u].x * lx;
          s.y = s.y + t[u].y * ly;
          if (g) {
            dvec2 o = {-lx, -ly};
            __builtin_nontemporal_store(o, reinterpret_cast<dvec2*>(g + (d0 + u) * ld + 2 * c2));
          }
        }
    }
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c2 < C2) {
    dvec2 tot = part[0][lane];
#pragma unroll
    for (int k = 1; k < RED_WAVES; ++k) {
      tot.x = tot.x + part[k][lane].x;
      tot.y = tot.y + part[k][lane].y;
    }
    tot.x = -0.5 * tot.x;
    tot.y = -0.5 * tot.y;
    *reinterpret_cast<dvec2*>(logp + 2 * c2) = tot;
  }
}

// The separable Gaussians as a per-coordinate term (bk_elementwise.hpp): term = theta*(lam*theta) summed, log p = -1/2 sum.
template <bool HL>
struct GaussTerm {
  __device__ __forceinline__ static void eval(double th, i64 d, const double* lam, double& term, double& grad) {
    const double lt = HL ? lam[d] * th : th;
    term = th * lt;
    grad = -lt;
  }
  __device__ __forceinline__ static double finish(double s) { return -0.5 * s; }
};

// Neal's funnel as a lane-spread density (bk_lanes.hpp): v = theta_0 is the head coordinate, the rows d >= 1 are
// exchangeable given v; the only coupling between the coordinates of a chain is s = sum_{d>=1} theta_d^2, summed in the
// canonical order of bk_lanes.hpp (16 interleaved class sums -> 4 group sums -> total, for every D).  Every funnel entry
// point -- gradient op (k_lane_op), one-launch leapfrog step, one-launch delayed-rejection proposal (k_lane_traj) -- is an
// instantiation of the library's templates with this density; a CTarget.from_source(form="lanes") density goes through
// the same templates.
struct FunnelDensity {
  static constexpr int HEAD = 1;
  template <class L>
  __device__ __forceinline__ static double eval(L& c, const double*) {
    const double v = c.head(0);
    const double s = c.sum([](double x, i64) { return x * x; });
    const double ev = bk_exp(-v);  // (the library's own exp: include/bkhip_math.h -- the same double on host and device)
    const double hn = 0.5 * (double)(c.dims() - 1);
    const double he = 0.5 * ev;
    c.grad_head(0, ((-v / 9.0) - hn) + he * s);
    c.grad([ev](double x, i64) { return -(ev * x); });
    return ((-(v * v) / 18.0) - hn * v) - he * s;
  }
};

int gauss(const double* theta, double* grad, double* logp, i64 ld, const double* lam, i64 C, i64 D,
          const uint32_t* n_dev, void* stream) {
  if (!theta || (!grad && !logp) || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (n_dev) {  // lane count on the device: one chain per lane, launch sized for the bound C
    if (logp)
      k_gauss_logp<<<dim3((unsigned)bk_cdiv(C, BK_WAVE)), dim3(RED_BLOCK), 0, s>>>(theta, grad, logp, ld, lam, C, D, n_dev);
    else if (D > 0)
      k_gauss_grad_s<<<dim3((unsigned)bk_cdiv(C, TG_BLOCK), (unsigned)bk_cdiv(D, TG_ROWS)), dim3(TG_BLOCK), 0, s>>>(
          theta, grad, ld, lam, C, D, n_dev);
    BK_RETURN_LAUNCH_STATUS();
  }
  if (logp) {
    if (C % 2 == 0 && ld % 2 == 0 && C * D >= ((i64)1 << 22) && bk_aligned16(theta) && bk_aligned16(logp) &&
        (!grad || bk_aligned16(grad)))
      k_gauss_logp_v2<<<dim3((unsigned)bk_cdiv(C / 2, BK_WAVE)), dim3(RED_BLOCK), 0, s>>>(theta, grad, logp, ld, lam,
                                                                                          C / 2, D);
    else
      k_gauss_logp<<<dim3((unsigned)bk_cdiv(C, BK_WAVE)), dim3(RED_BLOCK), 0, s>>>(theta, grad, logp, ld, lam, C, D,
                                                                                   nullptr);
    BK_RETURN_LAUNCH_STATUS();
  }
  if (D == 0) return BK_OK;
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(theta) && bk_aligned16(grad)) {
    static const int forced = []() { const char* e = getenv("BK_GG_VARIANT"); return e ? atoi(e) : -1; }();
    if (forced >= 0) {  // tuning only
#define BK_GG(ROWS, NT)                                                                      \
  do {                                                                                       \
    dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)bk_cdiv(D, ROWS));                \
    k_gauss_grad_v2<ROWS, NT><<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C / 2, D); \
  } while (0)
      switch (forced) {
        case 1: BK_GG(1, true); break;
        case 2: BK_GG(2, true); break;
        case 3: BK_GG(8, true); break;
        case 4: BK_GG(4, false); break;
        default: BK_GG(4, true); break;
      }
#undef BK_GG
    } else if (bk_streams_past_llc(2 * C * D) && D <= 65535) {
      // one row per thread, non-temporal: 6.5 TB/s vs 6.1 with four rows (MI355X, 1 GiB streams)
      dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)D);
      k_gauss_grad_v2<1, true><<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C / 2, D);
    } else {
      dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)bk_cdiv(D, 2));
      k_gauss_grad_v2<2, false><<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C / 2, D);
    }
  } else {
    dim3 grid((unsigned)bk_cdiv(C, TG_BLOCK), (unsigned)bk_cdiv(D, TG_ROWS));
    k_gauss_grad_s<<<grid, dim3(TG_BLOCK), 0, s>>>(theta, grad, ld, lam, C, D, nullptr);
  }
  BK_RETURN_LAUNCH_STATUS();
}

}  // namespace

extern "C" {

int bk_target_iso_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                                int64_t D, void* stream) {
  return gauss(theta, grad, logp, ld, nullptr, C, D, nullptr, stream);
}

int bk_target_diag_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld,
                                 const double* lam, int64_t C, int64_t D, void* stream) {
  if (!lam) return BK_E_ARG;
  return gauss(theta, grad, logp, ld, lam, C, D, nullptr, stream);
}

int bk_target_iso_gaussian_grad_n(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                                  int64_t D, const uint32_t* n_dev, void* stream) {
  return gauss(theta, grad, logp, ld, nullptr, C, D, n_dev, stream);
}

int bk_target_diag_gaussian_grad_n(const double* theta, double* grad, double* logp, int64_t ld,
                                   const double* lam, int64_t C, int64_t D, const uint32_t* n_dev, void* stream) {
  if (!lam) return BK_E_ARG;
  return gauss(theta, grad, logp, ld, lam, C, D, n_dev, stream);
}

int bk_hmc_trajectory_gaussian(const double* theta_in, double* theta_out, const double* rho_in,
                               double* rho_out, int64_t ld, const double* lam, const double* metric,
                               double eps, int64_t steps, int64_t C, int64_t D, void* stream) {
  // the library's whole-trajectory kernels (bk_elementwise.hpp) with the Gaussian term inlined
  if (lam)
    return bke::hmc_trajectory_launch<GaussTerm<true>>(theta_in, theta_out, rho_in, rho_out, ld, lam, metric, eps, steps, C, D,
                                                       stream);
  return bke::hmc_trajectory_launch<GaussTerm<false>>(theta_in, theta_out, rho_in, rho_out, ld, lam, metric, eps, steps, C, D,
                                                      stream);
}

int bk_hmc_draw_gaussian(const double* theta_in, double* theta_out, int64_t ld, const double* rho_in,
                         const double* zt, int64_t ldz, const double* lam, const double* metric, double eps,
                         int64_t steps, double* part, double* kin0, double* kin1, double* lp_out, double* lp_cur,
                         const double* log_u, uint8_t* accept_mask, double* ret, uint32_t* accept_count, int64_t C,
                         int64_t D, void* stream) {
  // the library's whole-draw kernels (bk_elementwise.hpp) with the Gaussian term inlined
  if (lam)
    return bke::hmc_draw_launch<GaussTerm<true>>(theta_in, theta_out, ld, rho_in, zt, ldz, lam, metric, eps, steps, part, kin0,
                                                 kin1, lp_out, lp_cur, log_u, accept_mask, ret, accept_count, C, D, stream);
  return bke::hmc_draw_launch<GaussTerm<false>>(theta_in, theta_out, ld, rho_in, zt, ldz, lam, metric, eps, steps, part, kin0,
                                                kin1, lp_out, lp_cur, log_u, accept_mask, ret, accept_count, C, D, stream);
}

int bk_mala_step_gaussian(const double* theta, double* theta_out, double* theta_prop, int64_t ld, const double* lam,
                          double* lp, const double* lp_prop, const double* log_u, const double* zt_next, int64_t ldz,
                          double eps, double sqrt2eps, uint8_t* accept_mask, double* ret, uint32_t* accept_count, int64_t C,
                          int64_t D, void* stream) {
  // the library's MALA step kernel for separable densities (bk_mala_step.hpp) with the Gaussian term inlined
  if (lam)
    return bkm::mala_step_sep_launch<GaussTerm<true>>(theta, theta_out, theta_prop, ld, lam, lp, lp_prop, log_u, zt_next, ldz, eps,
                                                      sqrt2eps, accept_mask, ret, accept_count, C, D, stream);
  return bkm::mala_step_sep_launch<GaussTerm<false>>(theta, theta_out, theta_prop, ld, lam, lp, lp_prop, log_u, zt_next, ldz, eps,
                                                     sqrt2eps, accept_mask, ret, accept_count, C, D, stream);
}

int bk_target_funnel_grad_n(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                            int64_t D, const uint32_t* n_dev, void* stream) {
  return bkl::target_launch<FunnelDensity>(theta, grad, logp, ld, nullptr, C, D, n_dev, stream);
}

int bk_target_funnel_grad(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                          int64_t D, void* stream) {
  return bkl::target_launch<FunnelDensity>(theta, grad, logp, ld, nullptr, C, D, nullptr, stream);
}

int bk_leapfrog_step_funnel(double* theta, double* rho, int64_t ld, const double* metric, double h, int64_t n, int64_t D,
                            const uint32_t* n_dev, void* stream) {
  return bkl::step_launch<FunnelDensity>(theta, rho, ld, metric, h, nullptr, n, D, n_dev, stream);
}

int bk_hmc_trajectory_funnel(const double* theta_in, double* rho, const double* grad_in, double* theta_out, double* grad_out,
                             double* logp_out, double* kin_out, int64_t ld, const double* metric, double eps, int64_t steps,
                             int64_t C, int64_t D, void* stream) {
  return bkl::hmc_trajectory_launch<FunnelDensity>(theta_in, rho, grad_in, ld, theta_out, grad_out, logp_out, kin_out, ld, metric,
                                                   eps, steps, C, D, nullptr, stream);
}

int bk_dr_proposal_funnel(const double* theta_in, const double* rho_in, const double* grad_in, int64_t ld_in,
                              const int32_t* src_index, double* theta_out, double* rho_out, double* grad_out,
                              double* logp_out, double* kin_out, int64_t ld_out, const double* metric, double h,
                              int64_t steps, int64_t n, int64_t D, const uint32_t* n_dev, uint32_t* lanes_out,
                              uint64_t* lanes_total, double* H_out, double* h_out, uint8_t* live_out,
                              const bk_scatter_job* job_in, const bk_ghost_link* ghost_in, const bk_ghost0* g0_in,
                              void* stream) {
  return bkl::dr_proposal_launch<FunnelDensity>(theta_in, rho_in, grad_in, ld_in, src_index, theta_out, rho_out, grad_out,
                                                logp_out, kin_out, ld_out, metric, h, steps, n, D, n_dev, lanes_out,
                                                lanes_total, H_out, h_out, live_out, job_in, ghost_in, g0_in, nullptr, stream);
}

}  // extern "C"
