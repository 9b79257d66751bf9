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
#include <stdlib.h>

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

// Neal's funnel, any D: one lane per chain, sequential in d
__global__ __launch_bounds__(PC_BLOCK) void k_funnel(const double* th, double* g, double* logp, i64 ld,
                                                     i64 C_host, i64 D, const uint32_t* n_dev) {
  const i64 C = bk_lanes(C_host, n_dev);
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

// Neal's funnel.  The only coupling between the coordinates of a chain is s = sum_{d>=1}
// theta_d^2, so the ROWS of a chain are split over lanes and s is reduced across them.
//
// Canonical summation order (every funnel kernel below; results do not depend on how many chains
// are in flight, on the grid, or on which geometry runs).  Row d belongs to class c = (d-1) mod 16,
// slot i = (d-1) div 16:
//     cs[c] = sum over i, in order, of x[1 + c + 16 i]^2              (16 class sums)
//     q[g]  = ((cs[g] + cs[g+4]) + cs[g+8]) + cs[g+12],  g = 0..3     (4 group sums)
//     s     = ((q[0] + q[1]) + q[2]) + q[3]
// Geometries that produce exactly these values:
//   * the gradient op (k_funnel_coop): a 4-wavefront workgroup, 64 chains, lane = chain; wavefront w owns
//     the four classes of group w and contributes q[w] through LDS.
//   * the trajectory kernel (k_funnel_traj): LPC adjacent lanes of ONE wavefront serve a chain, and s is
//     reduced with DPP moves inside the wavefront -- no LDS, no workgroup barrier in the leapfrog loop
//     (round 2 reduced through LDS with a 4-wavefront barrier per step: 0.59 us per step).
//       LPC = 4  (throughput: the first stage, every chain): 16 chains per wavefront; lane p of a chain's
//                quad holds the classes of group p (p, p+4, p+8, p+12: up to 32 rows), forms q[p] by itself,
//                and s is ((q0 + q1) + q2) + q3 over the quad (four quad_perm broadcasts).
//       LPC = 16 (latency: the sparse, long later stages -- a few per cent of the chains, 40 / 160 steps):
//                4 chains per wavefront, one 16-lane DPP row each; lane p holds class p (up to 8 rows).
//                Lanes 0..3 of the row fetch cs[p+4], cs[p+8], cs[p+12] with row_shl, form q[p], the quad
//                sums it to s, and two masked row_shr moves hand s to the other twelve lanes.
// Every lane integrates v = theta_0 redundantly (it needs exp(-v) for its own rows); one lane per
// chain writes it.
constexpr int FN_WAVES = 4;
constexpr int FN_CLASSES = 16;
constexpr int FN_MAX_SLOTS = 8;  // slots per class held in registers: D - 1 <= 128
constexpr int FN_MAX_ROWS = FN_CLASSES * FN_MAX_SLOTS;
constexpr int FN_BLOCK = FN_WAVES * BK_WAVE;

struct FunnelLds {
  double part[FN_WAVES][BK_WAVE];
};

// Geometry of the gradient op: register slot u = k*SL + i holds class w + 4k, slot i; chain = lane.
template <int SL>
struct FunnelGeo {
  static constexpr int KC = 4;        // classes per lane
  static constexpr int NU = KC * SL;  // register slots per lane
  __device__ static __forceinline__ i64 row(int w, int lane, int u) {
    return 1 + (w + 4 * (u / SL)) + (i64)FN_CLASSES * (u % SL);
  }
};

// s (canonical order) for this lane's chain from the lane's class sums cs[4] (one use per launch)
__device__ __forceinline__ double funnel_reduce_lds(FunnelLds& lds, int w, int lane, const double* cs) {
  lds.part[w][lane] = ((cs[0] + cs[1]) + cs[2]) + cs[3];
  __syncthreads();
  return ((lds.part[0][lane] + lds.part[1][lane]) + lds.part[2][lane]) + lds.part[3][lane];
}

// class sums of `expr` over this lane's rows < D, each class sequential in its slots
// (G::row(w, lane, u) with u = k*SL + i is the row of class slot (k, i))
#define BK_FN_CLASS_SUMS(cs, expr)                                   \
  _Pragma("unroll") for (int k = 0; k < G::KC; ++k) {                \
    double acc_ = 0.0;                                               \
    _Pragma("unroll") for (int i = 0; i < SL; ++i) {                 \
      const int u = k * SL + i;                                      \
      if (G::row(w, lane, u) < D) acc_ = acc_ + (expr);              \
    }                                                                \
    cs[k] = acc_;                                                    \
  }

// gradient / log density for n chains (chain j in column j)
template <int SL>
__global__ __launch_bounds__(FN_BLOCK) void k_funnel_coop(const double* th, double* g, double* logp, i64 ld,
                                                          i64 n_host, i64 D, const uint32_t* n_dev) {
  using G = FunnelGeo<SL>;
  __shared__ FunnelLds lds;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 n = bk_lanes(n_host, n_dev);
  if ((i64)blockIdx.x * BK_WAVE >= n) return;  // (whole workgroup past the set: uniform)
  const i64 j = (i64)blockIdx.x * BK_WAVE + lane;
  const bool on = j < n;
  double x[G::NU];
#pragma unroll
  for (int u = 0; u < G::NU; ++u) {
    const i64 d = G::row(w, lane, u);
    x[u] = (on && d < D) ? th[d * ld + j] : 0.0;
  }
  double cs[G::KC];
  BK_FN_CLASS_SUMS(cs, x[u] * x[u])
  double v = on ? th[j] : 0.0;
  double s = funnel_reduce_lds(lds, w, lane, cs);
  if (!on) return;
  double ev = exp(-v);
  double hn = 0.5 * (double)(D - 1);
  double he = 0.5 * ev;
  if (w == 0) {
    if (logp) logp[j] = ((-(v * v) / 18.0) - hn * v) - he * s;
    if (g) g[j] = ((-v / 9.0) - hn) + he * s;
  }
  if (g) {
#pragma unroll
    for (int u = 0; u < G::NU; ++u) {
      const i64 d = G::row(w, lane, u);
      if (d < D) g[d * ld + j] = -(ev * x[u]);
    }
  }
}

// ---- lanes of one wavefront serving one chain: geometry and the DPP reduction ------------------------
// DPP move of a double (two 32-bit halves).  Lanes the control does not reach (row / bank masks, a shift
// whose source lies outside the 16-lane row) keep `old`.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double bk_dpp_f64(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
constexpr int BK_DPP_ROW_SHL = 0x100;  // + n: lane i reads lane i + n of its row
constexpr int BK_DPP_ROW_SHR = 0x110;  // + n: lane i reads lane i - n of its row
// the same for controls under which every lane that matters has a source lane: no `old` operand, so no copy
// of the source in front of the move (lanes without a source read 0)
template <int CTRL>
__device__ __forceinline__ double bk_dpp_f64_all(double src) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(src), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(src), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ double bk_quad_bcast(double x) {  // lane K of every quad, to the whole quad
  return bk_dpp_f64_all<K * 0x55>(x);
}

//   LPC = 4 : p = lane % 4 is the class GROUP; register slot u = k*SL + i holds class p + 4k, slot i.
//   LPC = 8 : p = lane % 8;                    register slot u = k*SL + i holds class p + 8k, slot i (k = 0, 1).
//   LPC = 16: p = lane % 16 is the CLASS;      register slot u = i     holds class p,      slot i.
// A lane's rows are 1 + p + off(u) with a lane-independent off(u): row addresses are a per-lane 32-bit
// offset (row 1 + p of the lane's chain) on top of a wavefront-uniform row base, and only the LAST slot
// of a class can run past D (SL is exactly ceil((D-1)/16)): every other row needs no guard.
template <int LPC, int SL>
struct FunnelLanes {
  static_assert(LPC == 4 || LPC == 8 || LPC == 16, "a chain is served by a quad, half a DPP row or a DPP row");
  static constexpr int KC = FN_CLASSES / LPC;     // classes per lane
  static constexpr int NU = KC * SL;              // register slots per lane
  static constexpr int CHAINS = BK_WAVE / LPC;    // chains per wavefront
  __host__ __device__ static constexpr int off(int u) {  // row of slot u, relative to the lane's first row
    return LPC * (u / SL) + FN_CLASSES * (u % SL);  // (LPC = 16: one class per lane, u / SL = 0)
  }
  __host__ __device__ static constexpr bool last_slot(int u) { return u % SL == SL - 1; }
};

// s (canonical order) of the chain this lane serves, from the lane's class sums; valid in EVERY lane
// of the chain.  All 64 lanes must be active.
template <int LPC>
__device__ __forceinline__ double funnel_reduce_lanes(const double* cs) {
  double q;
  if (LPC == 4) {
    q = ((cs[0] + cs[1]) + cs[2]) + cs[3];  // this lane holds classes p, p+4, p+8, p+12
  } else if (LPC == 8) {
    // lane p of the 8-lane group holds cs[p] and cs[p+8]; lanes 0..3 fetch cs[p+4], cs[p+12] from lane p + 4
    const double b = bk_dpp_f64_all<BK_DPP_ROW_SHL + 4>(cs[0]);
    const double d = bk_dpp_f64_all<BK_DPP_ROW_SHL + 4>(cs[1]);
    q = ((cs[0] + b) + cs[1]) + d;
  } else {
    // lane p of the row holds cs[p]; in lanes 0..3: q[p] = ((cs[p] + cs[p+4]) + cs[p+8]) + cs[p+12]
    // (the other twelve lanes compute something nobody reads)
    const double b = bk_dpp_f64_all<BK_DPP_ROW_SHL + 4>(cs[0]);
    const double c = bk_dpp_f64_all<BK_DPP_ROW_SHL + 8>(cs[0]);
    const double d = bk_dpp_f64_all<BK_DPP_ROW_SHL + 12>(cs[0]);
    q = ((cs[0] + b) + c) + d;
  }
  // lane p of the quad holds q[p]
  double s = ((bk_quad_bcast<0>(q) + bk_quad_bcast<1>(q)) + bk_quad_bcast<2>(q)) + bk_quad_bcast<3>(q);
  if (LPC == 16) {
    // s is right in lanes 0..3 of the row: hand it to lanes 4..7, then lanes 0..7 hand it to 8..15
    s = bk_dpp_f64<BK_DPP_ROW_SHR + 4, 0xF, 0x2>(s, s);
    s = bk_dpp_f64<BK_DPP_ROW_SHR + 8, 0xF, 0xC>(s, s);
  } else if (LPC == 8) {
    // s is right in lanes 0..3 of each 8-lane group: hand it to lanes 4..7 (banks 1 and 3 of the row)
    s = bk_dpp_f64<BK_DPP_ROW_SHR + 4, 0xF, 0xA>(s, s);
  }
  return s;
}

// class sums of `expr` over this lane's rows (tail_ok[k]: the last slot of class k exists), each class
// sequential in its slots
#define BK_FL_CLASS_SUMS(cs, expr)                                   \
  _Pragma("unroll") for (int k = 0; k < G::KC; ++k) {                \
    double acc_ = 0.0;                                               \
    _Pragma("unroll") for (int i = 0; i < SL; ++i) {                 \
      const int u = k * SL + i;                                      \
      if (!G::last_slot(u) || tail_ok[k]) acc_ = acc_ + (expr);      \
    }                                                                \
    cs[k] = acc_;                                                    \
  }

// One whole delayed-rejection proposal (drghmc.py:319-346 -> :253-289) for n chains in ONE
// launch: gather chain idx[j] of the source point, first half-kick with the source's cached
// gradient + drift, (steps-1) x {gradient, kick, drift}, final gradient + log density,
// last half-kick, momentum flip, kinetic energy.  theta, rho never leave registers, and the sum over a
// chain's coordinates never leaves the wavefront.
// LPC: lanes per chain (above); SL = slots per class = ceil((D-1)/16) exactly; HM = a metric is given.
// All compile-time: with generic sizes and a run-time metric flag the kernel needed 330 registers (one
// wavefront per SIMD, AGPR spills).
// LPC_ARG = 4, 8 or 16: that geometry; 0: chosen at run time from the lane count, which only the device knows.
// A wavefront-step costs ~580 / ~700 / ~1100 cycles with 16 / 8 / 4 lanes per chain and serves 4 / 8 / 16
// chains; up to one wavefront per SIMD (1024 of them) the fewest cycles win, beyond that the fewest cycles per
// chain: 16 lanes per chain below FN_AUTO_MID lanes, 8 below FN_AUTO_WIDE, 4 from there on.  The grid is sized
// for the 16-lane form of the bound n_host.
constexpr i64 FN_AUTO_MID = 4608, FN_AUTO_WIDE = 12288;
template <int LPC, int SL, bool HM>
__device__ __forceinline__ void funnel_traj_body(
    const double* th_in, const double* rho_in, const double* g_in, i64 ld_in, const int32_t* idx,
    double* th_out, double* rho_out, double* g_out, double* logp_out, double* kin_out, i64 ld_out,
    const double* metric, double h, int steps, i64 n, i64 D, double* H_out, double* hh_out, uint8_t* live_out,
    const bk_ghost_link& ghost, const bk_ghost0& g0, int lane, int wave);

template <int LPC_ARG, int SL, bool HM>
__global__ __launch_bounds__(FN_BLOCK) void k_funnel_traj(
    const double* th_in, const double* rho_in, const double* g_in, i64 ld_in, const int32_t* idx,
    double* th_out, double* rho_out, double* g_out, double* logp_out, double* kin_out, i64 ld_out,
    const double* metric, double h, int steps, i64 n_host, i64 D, const uint32_t* n_dev, uint32_t* lanes_out,
    unsigned long long* lanes_total, double* H_out, double* hh_out, uint8_t* live_out, unsigned traj_blocks,
    bk_scatter_job job, bk_ghost_link ghost, bk_ghost0 g0) {
  const int lane = threadIdx.x & (BK_WAVE - 1), wave = bk_wave_id();
  if (blockIdx.x >= traj_blocks) {
    // surplus workgroups: the previous stage's scatter (bk_scatter_job), one 64-lane unit per wavefront
    i64 jn = job.n;
    if (job.n_dev) {
      const i64 m = (i64)*job.n_dev;
      jn = m < jn ? m : jn;
    }
    const i64 ux_count = (job.n + 63) / 64;  // (units are laid out for the host-side bound)
    const i64 unit = ((i64)blockIdx.x - traj_blocks) * FN_WAVES + wave;
    const i64 ux = unit % ux_count, uy = unit / ux_count;
    if (uy * BK_SCT_ROWS < job.D)
      bk_scatter_unit(ux, uy, lane, job.mask, job.index, jn, job.D, job.dst0, job.src0, job.dst1, job.src1, job.dst2,
                      job.src2, job.ld_dst, job.ld_src, job.sdst, job.ssrc);
    return;
  }
  // lanes actually in the set: read from device memory when the host only knows an upper bound
  i64 n = n_host;
  if (n_dev) {
    const i64 m = (i64)*n_dev;
    n = m < n ? m : n;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (lanes_out) *lanes_out = (uint32_t)n;
    if (lanes_total) *lanes_total += (unsigned long long)n;  // one writer per launch, launches are stream-ordered
    if (g0.steps > 0) {  // the fused first ghost runs over the same lanes
      if (g0.lanes_out) *g0.lanes_out = (uint32_t)n;
      if (g0.lanes_total) *reinterpret_cast<unsigned long long*>(g0.lanes_total) += (unsigned long long)n;
    }
  }
#define BK_FT_BODY(L)                                                                                              \
  funnel_traj_body<L, SL, HM>(th_in, rho_in, g_in, ld_in, idx, th_out, rho_out, g_out, logp_out, kin_out, ld_out, \
                              metric, h, steps, n, D, H_out, hh_out, live_out, ghost, g0, lane, wave)
  if (LPC_ARG == 4 || (LPC_ARG == 0 && n >= FN_AUTO_WIDE)) BK_FT_BODY(4);
  else if (LPC_ARG == 8 || (LPC_ARG == 0 && n >= FN_AUTO_MID)) BK_FT_BODY(8);
  else BK_FT_BODY(16);
#undef BK_FT_BODY
}

template <int LPC, int SL, bool HM>
__device__ __forceinline__ void funnel_traj_body(
    const double* th_in, const double* rho_in, const double* g_in, i64 ld_in, const int32_t* idx,
    double* th_out, double* rho_out, double* g_out, double* logp_out, double* kin_out, i64 ld_out,
    const double* metric, double h, int steps, i64 n, i64 D, double* H_out, double* hh_out, uint8_t* live_out,
    const bk_ghost_link& ghost, const bk_ghost0& g0, int lane, int wave) {
  using G = FunnelLanes<LPC, SL>;
  constexpr int NU = G::NU;
  const i64 j0 = ((i64)blockIdx.x * FN_WAVES + wave) * G::CHAINS;  // first chain of this wavefront
  if (j0 >= n) return;  // whole wavefront past the set (uniform; the wavefronts of a workgroup are independent)
  const int pos = lane & (LPC - 1);
  const i64 j = j0 + lane / LPC;
  const bool on = j < n;
  const bool writer = on && pos == 0;  // the lane that owns theta_0 / the scalars
  const i64 src = on ? (idx ? (i64)idx[j] : j) : 0;
  const double half = 0.5 * h;
  const double hn = 0.5 * (double)(D - 1);
  constexpr bool hm = HM;
  // this lane's rows: d(u) = 1 + pos + off(u).  Byte offsets of its first row (host checked: < 2^32)
  const uint32_t bo_in = (uint32_t)(((i64)(1 + pos) * ld_in + src) * 8);
  const uint32_t bo_out = (uint32_t)(((i64)(1 + pos) * ld_out + (on ? j : 0)) * 8);
#define BK_FL_IN(p, u) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>((p) + (i64)G::off(u) * ld_in) + bo_in))
#define BK_FL_OUT(p, u) (*reinterpret_cast<double*>(reinterpret_cast<char*>((p) + (i64)G::off(u) * ld_out) + bo_out))
  bool tail_ok[G::KC];  // does the last slot of class k exist for this lane?
#pragma unroll
  for (int k = 0; k < G::KC; ++k) tail_ok[k] = 1 + pos + G::off(k * SL + SL - 1) < D;
#define BK_FL_OK(u) (!G::last_slot(u) || tail_ok[(u) / SL])
  double x[NU], r[NU], mt[HM && LPC == 16 ? NU : 1];
  // gather + first half-kick + drift (drghmc.py:276-278)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    // (lanes past the set read chain 0 -- src = 0 -- and compute on it; they store nothing.  No branch
    // around the loads of rows that exist for every lane.)
    const bool ok = BK_FL_OK(u);
    x[u] = ok ? BK_FL_IN(th_in, u) : 0.0;
    r[u] = ok ? BK_FL_IN(rho_in, u) : 0.0;
    double g0 = ok ? BK_FL_IN(g_in, u) : 0.0;
    const double mi = (hm && BK_FL_OK(u)) ? metric[1 + pos + G::off(u)] : 1.0;
    if (HM && LPC == 16) mt[u] = mi;
    double t = hm ? mi * g0 : g0;
    r[u] = r[u] + half * t;
    x[u] = x[u] + h * r[u];
    // the gather is issued in batches of 8 rows: all 3*NU loads in flight at once would set the
    // kernel's register count (and so its occupancy for the whole trajectory)
    if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  double v = th_in[src], rv = rho_in[src];
  const double mv = hm ? metric[0] : 1.0;
  {
    double g0 = g_in[src];
    double t = hm ? mv * g0 : g0;
    rv = rv + half * t;
    v = v + h * rv;
  }
  // the metric entry of row u: held in registers by the 16-lane geometry (<= 8 rows), re-read (L1) by the quad one
#define BK_FL_METRIC(u) ((HM && LPC == 16) ? mt[(HM && LPC == 16) ? (u) : 0] : metric[1 + pos + G::off(u)])
  double cs[G::KC];
  // (steps-1) x {gradient, kick, drift} (drghmc.py:280-283)
  for (int step = 0; step + 1 < steps; ++step) {
    BK_FL_CLASS_SUMS(cs, x[u] * x[u])
    const double s = funnel_reduce_lanes<LPC>(cs);
    const double ev = exp(-v);
    const double gv = ((-v / 9.0) - hn) + (0.5 * ev) * s;
    {
      double t = hm ? mv * gv : gv;
      rv = rv + h * t;
      v = v + h * rv;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (BK_FL_OK(u)) {
        double gi = -(ev * x[u]);
        double t = hm ? BK_FL_METRIC(u) * gi : gi;
        r[u] = r[u] + h * t;
        x[u] = x[u] + h * r[u];
      }
      if (LPC != 16 && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the live temporaries (registers -> occupancy)
    }
  }
  // final gradient + log density (drghmc.py:285) and the last half-kick (:286)
  double logp_j = 0.0, ev_end = 0.0, gv_end = 0.0;
  {
    BK_FL_CLASS_SUMS(cs, x[u] * x[u])
    const double s = funnel_reduce_lanes<LPC>(cs);
    const double ev = exp(-v);
    const double he = 0.5 * ev;
    const double gv = ((-v / 9.0) - hn) + he * s;
    ev_end = ev;
    gv_end = gv;
    {
      double t = hm ? mv * gv : gv;
      rv = rv + half * t;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (BK_FL_OK(u)) {
        double gi = -(ev * x[u]);
        double t = hm ? BK_FL_METRIC(u) * gi : gi;
        r[u] = r[u] + half * t;
        if (on) BK_FL_OUT(g_out, u) = gi;
      }
    }
    logp_j = ((-(v * v) / 18.0) - hn * v) - he * s;
    if (writer) {
      g_out[j] = gv;
      logp_out[j] = logp_j;
    }
  }
  // momentum flip (drghmc.py:345), kinetic energy (drghmc.py:250; same canonical order), outputs
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (BK_FL_OK(u)) {
      r[u] = -r[u];
      if (on) {
        BK_FL_OUT(rho_out, u) = r[u];
        BK_FL_OUT(th_out, u) = x[u];
      }
    }
  }
  BK_FL_CLASS_SUMS(cs, r[u] * (hm ? BK_FL_METRIC(u) * r[u] : r[u]))
  double ksum = funnel_reduce_lanes<LPC>(cs);
  double H_own = 0.0;
  if (writer) {
    double rr = -rv;
    double mr = hm ? mv * rr : rr;
    rho_out[j] = rr;
    th_out[j] = v;
    const double kin = 0.5 * (rr * mr + ksum);
    kin_out[j] = kin;
    if (H_out) {
      // the level set-up of accept() (bk_dr_level_begin) for this lane, in the same launch:
      // H = -((-logp) + kin) (drghmc.py:421 -> :249-251); h and live follow below
      const double potential = -logp_j;
      const double Hj = -(potential + kin);
      H_own = Hj;
      H_out[j] = Hj;
    }
  }
  double h_own = 0.0;    // the produced level's h / live for this lane (h = 0, live = 1 unless its first ghost
  bool live_own = true;  // is run here as well)

  // ---- the proposal's FIRST GHOST, in the same wavefront (drghmc.py:424-436 with i = 0) ----------------------
  // Every lane of a level gets ghost 0, lane for lane, and a ghost of the first proposal kind has no ghosts of its
  // own: the wavefront that has just produced proposal P integrates P's ghost straight from its registers
  // (theta_P, the flipped momentum, the gradient at theta_P = -(e^-v x) with the e^-v of the last evaluation) --
  // no store + gather of the ghost's source, no ghost arrays at all (only its joint log density is ever used),
  // one launch instead of two.  Same operation sequence as a launch of its own.
  if (g0.steps > 0) {
    const double h2 = g0.h, half2 = 0.5 * g0.h;
    // first half-kick + drift from the proposal's end point (drghmc.py:276-278)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (BK_FL_OK(u)) {
        double gi = -(ev_end * x[u]);
        double t = hm ? BK_FL_METRIC(u) * gi : gi;
        r[u] = r[u] + half2 * t;
        x[u] = x[u] + h2 * r[u];
      }
    }
    double rv2 = -rv;
    {
      double t = hm ? mv * gv_end : gv_end;
      rv2 = rv2 + half2 * t;
      v = v + h2 * rv2;
    }
    for (int step = 0; step + 1 < g0.steps; ++step) {
      BK_FL_CLASS_SUMS(cs, x[u] * x[u])
      const double s = funnel_reduce_lanes<LPC>(cs);
      const double ev = exp(-v);
      const double gv = ((-v / 9.0) - hn) + (0.5 * ev) * s;
      {
        double t = hm ? mv * gv : gv;
        rv2 = rv2 + h2 * t;
        v = v + h2 * rv2;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (BK_FL_OK(u)) {
          double gi = -(ev * x[u]);
          double t = hm ? BK_FL_METRIC(u) * gi : gi;
          r[u] = r[u] + h2 * t;
          x[u] = x[u] + h2 * r[u];
        }
        if (LPC != 16 && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    double logp_g = 0.0;
    {
      BK_FL_CLASS_SUMS(cs, x[u] * x[u])
      const double s = funnel_reduce_lanes<LPC>(cs);
      const double ev = exp(-v);
      const double he = 0.5 * ev;
      const double gv = ((-v / 9.0) - hn) + he * s;
      {
        double t = hm ? mv * gv : gv;
        rv2 = rv2 + half2 * t;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (BK_FL_OK(u)) {
          double gi = -(ev * x[u]);
          double t = hm ? BK_FL_METRIC(u) * gi : gi;
          r[u] = r[u] + half2 * t;
          r[u] = -r[u];
        }
      }
      logp_g = ((-(v * v) / 18.0) - hn * v) - he * s;
    }
    BK_FL_CLASS_SUMS(cs, r[u] * (hm ? BK_FL_METRIC(u) * r[u] : r[u]))
    const double ksum_g = funnel_reduce_lanes<LPC>(cs);
    bool goes_on = false;
    if (writer) {
      const double rr = -rv2;
      const double mr = hm ? mv * rr : rr;
      const double kin_g = 0.5 * (rr * mr + ksum_g);
      const double H_g = -((-logp_g) + kin_g);
      // the ghost against the proposal it came from (parent h = 0: this is the proposal's first ghost), then the
      // proposal's level entry: bk_dr_level_begin + bk_dr_accept_prob_ghost in one
      const double g = dr_accept_logprob(H_g, H_own, 0.0, 0.0, g0.prob_retry);
      if (g == 0.0) {  // drghmc.py:430-432
        live_own = false;
        g0.parent_a[j] = -INFINITY;
      } else {
        h_own = 0.0 + log1p(-exp(g));  // drghmc.py:434-435
        goes_on = true;
      }
    }
    if (g0.next_index) bk_append(goes_on, (int32_t)j, g0.next_index, g0.next_count);
  }
  if (writer && H_out) {
    hh_out[j] = h_own;
    live_out[j] = live_own ? 1 : 0;
  }

  // ---- a GHOST level that is complete here (no ghosts of its own, or one and it ran above): its acceptance
  // probability against the parent lane it came from and the parent's update (bk_dr_accept_prob_ghost;
  // drghmc.py:426-446), instead of a launch of their own.  One ghost lane per parent lane: nobody else touches
  // lane `src` of the parent level.
  if (ghost.parent_H) {
    bool parent_goes_on = false;
    if (writer) {
      double g = -INFINITY;  // (a dead lane: one of its own ghosts was accepted with probability one)
      if (live_own) {
        g = dr_accept_logprob(H_own, ghost.parent_H[src], h_own, ghost.parent_h[src], ghost.prob_retry);
        ghost.a_out[j] = g;
      }
      if (g == 0.0) {  // drghmc.py:430-432
        ghost.parent_a[src] = -INFINITY;
        ghost.parent_live[src] = 0;
      } else {
        ghost.parent_h[src] = ghost.parent_h[src] + log1p(-exp(g));  // drghmc.py:434-435
        parent_goes_on = true;
      }
    }
    // the parent lanes that go on to their next ghost: the lane set of that trajectory (every lane takes part)
    if (ghost.next_index) bk_append(parent_goes_on, (int32_t)src, ghost.next_index, ghost.next_count);
  }
#undef BK_FL_IN
#undef BK_FL_OUT
#undef BK_FL_OK
#undef BK_FL_METRIC
}

// Whole HMC trajectory of the separable Gaussians in registers (hmc.py:40-53 with
// grad = -(lam*theta) inlined): every (d, c) element is independent of all others through
// the L steps, so theta and rho are read once and written once (32 B per element per
// TRAJECTORY instead of 56 B per element per STEP) and the kernel is bound by the fp64
// vector units, not by HBM.  The per-element operation sequence is exactly the one the
// step-by-step kernels execute, so the results are bit-identical.
// HL / HM: lam / metric given (compile-time, so that the loop carries no selects); the ROWS rows of
// a thread advance together, which gives 2*ROWS independent dependency chains per lane.
template <int ROWS, bool HL, bool HM>
__global__ __launch_bounds__(TG_BLOCK) void k_traj_gauss(const double* th_in, double* th_out,
                                                         const double* rho_in, double* rho_out, i64 ld,
                                                         const double* lam, const double* metric, double eps,
                                                         int steps, i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * TG_BLOCK + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  const double half = 0.5 * eps;
  dvec2 th[ROWS], r[ROWS], t[ROWS];
  double l[ROWS], m[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const i64 d = (d0 + i < D) ? d0 + i : D - 1;  // rows past the end recompute the last one, not stored
    th[i] = *reinterpret_cast<const dvec2*>(th_in + d * ld + 2 * c2);
    r[i] = *reinterpret_cast<const dvec2*>(rho_in + d * ld + 2 * c2);
    l[i] = HL ? lam[d] : 1.0;
    m[i] = HM ? metric[d] : 1.0;
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    double gx = HL ? -(l[i] * th[i].x) : -th[i].x, gy = HL ? -(l[i] * th[i].y) : -th[i].y;
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
      double gx = HL ? -(l[i] * th[i].x) : -th[i].x;  // hmc.py:50
      double gy = HL ? -(l[i] * th[i].y) : -th[i].y;
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

__global__ __launch_bounds__(TG_BLOCK) void k_traj_gauss_s(const double* th_in, double* th_out,
                                                           const double* rho_in, double* rho_out, i64 ld,
                                                           const double* lam, const double* metric, double eps,
                                                           int steps, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * TG_BLOCK + threadIdx.x;
  i64 d = blockIdx.y;
  if (c >= C) return;
  const double half = 0.5 * eps;
  double th = th_in[d * ld + c], r = rho_in[d * ld + c];
  const double l = lam ? lam[d] : 1.0, m = metric ? metric[d] : 1.0;
  double g = lam ? -(l * th) : -th;
  double t = metric ? m * g : g;
  r = r + (-half) * t;
  for (int n = 0; n < steps; ++n) {
    r = r + eps * t;
    th = th + eps * r;
    g = lam ? -(l * th) : -th;
    t = metric ? m * g : g;
  }
  r = r + half * t;
  th_out[d * ld + c] = th;
  rho_out[d * ld + c] = r;
}

// The same trajectory as the trajectory + energies of a whole HMC draw (hmc.py:55-59): one thread
// per (chain, QUARTER of the dimensions) walks its quarter in chunks of TQ_ROWS rows -- each chunk
// is a register-resident trajectory as above -- and accumulates, sequentially in d, the three
// per-chain sums a draw needs:
//     kin0 = 1/2 sum rho0*(m*rho0)   [hmc.py:57 -> :37]    (momentum as drawn)
//     kin1 = 1/2 sum rho1*(m*rho1)   [hmc.py:59 -> :37]    (momentum at the end)
//     lp1  = -1/2 sum theta1*(lam*theta1)                   (the target's log density at the end)
// as quarter partials part[k][q][c]; combined ((p0+p1)+p2)+p3 (k_quarter_sums) they are, bit for
// bit, what bk_leapfrog_finish and bk_target_*_gaussian_grad compute with their four wavefronts
// per 64 chains -- which is why the walk is per quarter and in d order.  The momentum is read
// either in the state layout (rho_in) or straight from the wavefront-per-chain generator's
// chain-major normals (zt[c*ldz + d], rho0 = 0.0 + 1.0*z as numpy's random_normal); the final
// momentum is never stored (HMC discards it).  HBM traffic: 8 D (theta) + 8 D (momentum) read,
// 8 D (theta') written per chain -- against 32 D for the trajectory kernel above plus 32 D for
// the two reductions it replaces.
constexpr int TQ_ROWS = 8;
template <bool HL, bool HM, bool ZT>
__global__ __launch_bounds__(TG_BLOCK) void k_traj_gauss_q(const double* th_in, double* th_out, const double* rho_in,
                                                           i64 ld, const double* zt, i64 ldz, const double* lam,
                                                           const double* metric, double eps, int steps,
                                                           double* part, i64 C, i64 D) {
  const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;  // (64-thread workgroups for small launches)
  const int q = blockIdx.y;
  if (c >= C) return;
  const i64 Dq = (D + 3) / 4;
  const i64 dlo = q * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  const double half = 0.5 * eps;
  double k0 = 0.0, k1 = 0.0, sl = 0.0;
  for (i64 d0 = dlo; d0 < dhi; d0 += TQ_ROWS) {
    double th[TQ_ROWS], r[TQ_ROWS], t[TQ_ROWS], l[TQ_ROWS], m[TQ_ROWS];
#pragma unroll
    for (int i = 0; i < TQ_ROWS; ++i) {
      const i64 d = (d0 + i < dhi) ? d0 + i : dhi - 1;  // rows past the end recompute the last one, unused
      th[i] = th_in[d * ld + c];
      r[i] = ZT ? 0.0 + 1.0 * zt[c * ldz + d] : rho_in[d * ld + c];
      l[i] = HL ? lam[d] : 1.0;
      m[i] = HM ? metric[d] : 1.0;
    }
#pragma unroll
    for (int i = 0; i < TQ_ROWS; ++i) {
      if (d0 + i < dhi) {
        const double mv = HM ? m[i] * r[i] : r[i];
        k0 = k0 + r[i] * mv;
      }
      const double g = HL ? -(l[i] * th[i]) : -th[i];
      t[i] = HM ? m[i] * g : g;
      r[i] = r[i] + (-half) * t[i];  // hmc.py:46
    }
    for (int n = 0; n < steps; ++n) {
#pragma unroll
      for (int i = 0; i < TQ_ROWS; ++i) {
        r[i] = r[i] + eps * t[i];    // hmc.py:48
        th[i] = th[i] + eps * r[i];  // hmc.py:49
        const double g = HL ? -(l[i] * th[i]) : -th[i];  // hmc.py:50
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
        const double lt = HL ? l[i] * th[i] : th[i];
        sl = sl + th[i] * lt;
      }
    }
  }
  part[(0 * 4 + q) * C + c] = k0;
  part[(1 * 4 + q) * C + c] = k1;
  part[(2 * 4 + q) * C + c] = sl;
}

// ... and, when the caller hands over the rest of the accept test's inputs, the test itself
// (hmc.py:60-63, the arithmetic of bk_mh_accept in HMC mode): one launch less per draw.
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
    const double a0 = 0.5 * v[0], a1 = 0.5 * v[1], l1 = -0.5 * v[2];
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
  if (!theta_in || !theta_out || !rho_in || !rho_out || steps < 0 || steps > 0x7fffffff || C < 0 || D < 0)
    return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(theta_in) && bk_aligned16(theta_out) && bk_aligned16(rho_in) &&
      bk_aligned16(rho_out)) {
    constexpr int TRAJ_ROWS = 4;  // 8 independent dependency chains per lane (1 / 2 / 4 rows: 1.01 / 0.93 / 0.90 ms)
    dim3 grid((unsigned)bk_cdiv(C / 2, TG_BLOCK), (unsigned)bk_cdiv(D, TRAJ_ROWS));
#define BK_TRAJ(HL, HM)                                                                                     \
  k_traj_gauss<TRAJ_ROWS, HL, HM><<<grid, dim3(TG_BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, lam, \
                                                                  metric, eps, (int)steps, C / 2, D)
    if (lam && metric) BK_TRAJ(true, true);
    else if (lam) BK_TRAJ(true, false);
    else if (metric) BK_TRAJ(false, true);
    else BK_TRAJ(false, false);
#undef BK_TRAJ
  } else {
    dim3 grid((unsigned)bk_cdiv(C, TG_BLOCK), (unsigned)D);
    k_traj_gauss_s<<<grid, dim3(TG_BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, lam, metric, eps,
                                                   (int)steps, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_hmc_draw_gaussian(const double* theta_in, double* theta_out, int64_t ld, const double* rho_in,
                         const double* zt, int64_t ldz, const double* lam, const double* metric, double eps,
                         int64_t steps, double* part, double* kin0, double* kin1, double* lp_out, double* lp_cur,
                         const double* log_u, uint8_t* accept_mask, double* ret, uint32_t* accept_count, int64_t C,
                         int64_t D, void* stream) {
  if (!theta_in || !theta_out || (!rho_in && !zt) || (rho_in && zt) || !part || !kin1 || !lp_out || steps < 0 ||
      steps > 0x7fffffff || C < 0 || D < 0 || (lp_cur && !log_u))
    return BK_E_ARG;
  if (ld < C || (zt && ldz < D)) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  // one wavefront per workgroup while that still leaves CUs idle (4096 chains: 256 workgroups instead of 64)
  const int tq_block = C * 4 <= 256 * TG_BLOCK ? BK_WAVE : TG_BLOCK;
  dim3 grid((unsigned)bk_cdiv(C, tq_block), 4);
#define BK_TQ(HL, HM)                                                                                               \
  do {                                                                                                              \
    if (zt)                                                                                                         \
      k_traj_gauss_q<HL, HM, true><<<grid, dim3(tq_block), 0, s>>>(theta_in, theta_out, rho_in, ld, zt, ldz, lam,    \
                                                                   metric, eps, (int)steps, part, C, D);            \
    else                                                                                                            \
      k_traj_gauss_q<HL, HM, false><<<grid, dim3(tq_block), 0, s>>>(theta_in, theta_out, rho_in, ld, zt, ldz, lam,   \
                                                                    metric, eps, (int)steps, part, C, D);           \
  } while (0)
  if (lam && metric) BK_TQ(true, true);
  else if (lam) BK_TQ(true, false);
  else if (metric) BK_TQ(false, true);
  else BK_TQ(false, false);
#undef BK_TQ
  k_quarter_sums<<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, s>>>(part, kin0, kin1, lp_out, lp_cur, log_u,
                                                                       accept_mask, ret, accept_count, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_target_funnel_grad_n(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                            int64_t D, const uint32_t* n_dev, void* stream) {
  if (!theta || (!grad && !logp) || C < 0 || D < 1) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  if (D - 1 <= FN_MAX_ROWS) {
    const int need = (int)((D - 1 + FN_CLASSES - 1) / FN_CLASSES);
    dim3 grid((unsigned)bk_cdiv(C, BK_WAVE)), block(FN_BLOCK);
    hipStream_t s = bk_stream(stream);
    if (need <= 2) k_funnel_coop<2><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
    else if (need <= 4) k_funnel_coop<4><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
    else if (need <= 7) k_funnel_coop<7><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
    else k_funnel_coop<8><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
  }
  else
    k_funnel<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(theta, grad, logp,
                                                                                           ld, C, D, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_target_funnel_grad(const double* theta, double* grad, double* logp, int64_t ld, int64_t C,
                          int64_t D, void* stream) {
  return bk_target_funnel_grad_n(theta, grad, logp, ld, C, D, nullptr, stream);
}

int bk_dr_proposal_funnel_job(const double* theta_in, const double* rho_in, const double* grad_in, int64_t ld_in,
                              const int32_t* src_index, double* theta_out, double* rho_out, double* grad_out,
                              double* logp_out, double* kin_out, int64_t ld_out, const double* metric, double h,
                              int64_t steps, int64_t n, int64_t D, const uint32_t* n_dev, uint32_t* lanes_out,
                              uint64_t* lanes_total, double* H_out, double* h_out, uint8_t* live_out,
                              const bk_scatter_job* job_in, const bk_ghost_link* ghost_in, const bk_ghost0* g0_in,
                              void* stream) {
  if (!theta_in || !rho_in || !grad_in || !theta_out || !rho_out || !grad_out || !logp_out || !kin_out ||
      steps < 1 || steps > 0x7fffffff || n < 0 || D < 1)
    return BK_E_ARG;
  if (D - 1 > FN_MAX_ROWS) return BK_E_ARG;  // caller falls back to the step-by-step path
  if (H_out && (!h_out || !live_out)) return BK_E_ARG;
  if (ld_out < n) return BK_E_ALIGN;
  bk_ghost_link ghost = {};
  if (ghost_in) {
    ghost = *ghost_in;
    if (!H_out || !ghost.parent_H || !ghost.parent_h || !ghost.parent_live || !ghost.parent_a || !ghost.a_out ||
        (ghost.next_index && (!ghost.next_count || ghost.next_index == src_index)))
      return BK_E_ARG;
  }
  bk_ghost0 g0 = {};
  if (g0_in) {
    g0 = *g0_in;
    // (a level that has further ghosts is not complete in this launch: no link)
    if (g0.steps < 1 || g0.steps > 0x7fffffff || !H_out || !g0.parent_a ||
        (g0.next_index && (!g0.next_count || g0.next_index == src_index || ghost_in)))
      return BK_E_ARG;
  }
  bk_scatter_job job = {};
  unsigned job_blocks = 0;
  if (job_in) {
    job = *job_in;
    if (!job.mask || !job.dst0 || !job.src0 || (job.dst1 && !job.src1) || (job.dst2 && !job.src2) ||
        (job.sdst && !job.ssrc) || job.n < 0 || job.D < 0)
      return BK_E_ARG;
    if (job.n > 0 && job.D > 0)
      job_blocks = (unsigned)bk_cdiv(bk_cdiv(job.n, 64) * bk_cdiv(job.D, BK_SCT_ROWS), FN_WAVES);
  }
  // the kernel addresses a lane's rows with 32-bit byte offsets from wavefront-uniform row bases
  if ((ld_in > ld_out ? ld_in : ld_out) >= ((i64)1 << 32) / (8 * (FN_CLASSES + 1))) return BK_E_ARG;
  if (n == 0) {
    if (job_blocks) {  // nothing to propose: the job still runs
      int rc = bk_scatter_columns(job.mask, job.index, job.n, job.D, job.dst0, job.src0, job.dst1, job.src1, job.dst2,
                                  job.src2, job.ld_dst, job.ld_src, job.sdst, job.ssrc, job.n_dev, stream);
      if (rc != BK_OK) return rc;
    }
    if (g0.steps > 0 && g0.lanes_out) {
      int rc = (int)hipMemsetAsync(g0.lanes_out, 0, sizeof(uint32_t), bk_stream(stream));
      if (rc != 0) return rc;
    }
    if (lanes_out) return (int)hipMemsetAsync(lanes_out, 0, sizeof(uint32_t), bk_stream(stream));
    return BK_OK;
  }
  const int need = (int)((D - 1 + FN_CLASSES - 1) / FN_CLASSES);
  hipStream_t s = bk_stream(stream);
  // Geometry.  A set known to be small -> 16 lanes per chain, 16 chains per workgroup; known to be large -> 4
  // lanes per chain, 64 chains per workgroup; a set whose size only the device knows (every set after the first
  // stage) -> decided by the kernel from *n_dev when the bound allows a large one.  Same values either way.
  // BK_FUNNEL_GEOMETRY=wide|narrow overrides (wide = 4 lanes per chain).
  static const int forced = []() {
    const char* e = getenv("BK_FUNNEL_GEOMETRY");
    return !e ? 0 : (e[0] == 'n' ? 2 : e[0] == 'm' ? 3 : 1);  // wide | mid (8 lanes per chain) | narrow
  }();
  // 16: 16 lanes per chain; 4: 4 lanes per chain; 0: the kernel decides from *n_dev
  int geo;
  if (forced) geo = forced == 2 ? 16 : (forced == 3 ? 8 : 4);
  else if (n_dev) geo = n >= FN_AUTO_MID ? 0 : 16;
  else geo = n >= FN_AUTO_WIDE ? 4 : (n >= FN_AUTO_MID ? 8 : 16);
  const unsigned traj_blocks = (unsigned)bk_cdiv(n, FN_WAVES * (geo == 4 ? BK_WAVE / 4 : geo == 8 ? BK_WAVE / 8 : BK_WAVE / 16));
  dim3 grid(traj_blocks + job_blocks);
#define BK_FT(LPC, R, M)                                                                                          \
  k_funnel_traj<LPC, R, M><<<grid, dim3(FN_BLOCK), 0, s>>>(theta_in, rho_in, grad_in, ld_in, src_index, theta_out, \
                                                           rho_out, grad_out, logp_out, kin_out, ld_out, metric, h, \
                                                           (int)steps, n, D, n_dev, lanes_out,                     \
                                                           reinterpret_cast<unsigned long long*>(lanes_total),    \
                                                           H_out, h_out, live_out, traj_blocks, job, ghost, g0)
#define BK_FT_ROWS(R)                   \
  do {                                  \
    if (geo == 16) {                    \
      if (metric) BK_FT(16, R, true);   \
      else BK_FT(16, R, false);         \
    } else if (geo == 4) {              \
      if (metric) BK_FT(4, R, true);    \
      else BK_FT(4, R, false);          \
    } else if (geo == 8) {              \
      if (metric) BK_FT(8, R, true);    \
      else BK_FT(8, R, false);          \
    } else {                            \
      if (metric) BK_FT(0, R, true);    \
      else BK_FT(0, R, false);          \
    }                                   \
  } while (0)
  switch (need < 1 ? 1 : need) {  // slots per class, exactly: only a class's last slot can run past D
    case 1: BK_FT_ROWS(1); break;
    case 2: BK_FT_ROWS(2); break;
    case 3: BK_FT_ROWS(3); break;
    case 4: BK_FT_ROWS(4); break;
    case 5: BK_FT_ROWS(5); break;
    case 6: BK_FT_ROWS(6); break;
    case 7: BK_FT_ROWS(7); break;
    default: BK_FT_ROWS(8); break;
  }
#undef BK_FT_ROWS
#undef BK_FT
  BK_RETURN_LAUNCH_STATUS();
}


int bk_dr_proposal_funnel(const double* theta_in, const double* rho_in, const double* grad_in, int64_t ld_in,
                          const int32_t* src_index, double* theta_out, double* rho_out, double* grad_out,
                          double* logp_out, double* kin_out, int64_t ld_out, const double* metric, double h,
                          int64_t steps, int64_t n, int64_t D, const uint32_t* n_dev, uint32_t* lanes_out,
                          uint64_t* lanes_total, double* H_out, double* h_out, uint8_t* live_out, void* stream) {
  return bk_dr_proposal_funnel_job(theta_in, rho_in, grad_in, ld_in, src_index, theta_out, rho_out, grad_out, logp_out,
                                   kin_out, ld_out, metric, h, steps, n, D, n_dev, lanes_out, lanes_total, H_out, h_out,
                                   live_out, nullptr, nullptr, nullptr, stream);
}

}  // extern "C"
