// Leapfrog integrator, accept/select and MALA proposal-density kernels.
//
// Layout: phase-space arrays are [D][ld] with the chain index contiguous, one chain per
// lane.  Elementwise kernels (kick+drift, select) tile the (d, c) plane so that every
// wavefront moves 64 lanes x 16 B = 1 KiB per instruction; per-chain reductions (kinetic
// energy, proposal densities) run one lane per chain sequentially over d, which keeps the
// summation order fixed (deterministic, independent of the grid) and needs no cross-lane
// traffic.  The library is compiled with -ffp-contract=off: no FMA, every product and sum
// is rounded separately, exactly like the NumPy expressions of the reference.
#include "bk_common.hpp"
#include <stdlib.h>

namespace {

// ---- fused kick + drift ---------------------------------------------------------------
constexpr int KD_ROWS = 4;     // rows (dimensions) per thread: 12 x 16-B loads in flight per lane
constexpr int KD_BLOCK = 256;  // 4 wavefronts

__device__ __forceinline__ double kd_elem(double th, double rho, double g, double m, bool has_m,
                                          double eps, int use_pre, double pre, int use_kick,
                                          double kick, double& rho_new) {
  double t = has_m ? m * g : g;  // metric NULL == ones: 1.0*g == g exactly
  double r = rho;
  if (use_pre) r = r + pre * t;
  if (use_kick) r = r + kick * t;
  rho_new = r;
  return th + eps * r;
}

// chain-contiguous gradient, two chains (16 B) per lane
typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int ROWS, bool NT>
__global__ __launch_bounds__(KD_BLOCK) void k_kick_drift_v2(
    const double* th_in, double* th_out, const double* rho_in, double* rho_out, i64 ld,
    const double* grad, i64 ldg, const double* metric, double eps, int use_pre, double pre,
    int use_kick, double kick, i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * KD_BLOCK + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  dvec2 t[ROWS], r[ROWS], g[ROWS];
  double m[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    i64 d = d0 + i;
    if (d < D) {
      const dvec2* pt = reinterpret_cast<const dvec2*>(th_in + d * ld + 2 * c2);
      const dvec2* pr = reinterpret_cast<const dvec2*>(rho_in + d * ld + 2 * c2);
      const dvec2* pg = reinterpret_cast<const dvec2*>(grad + d * ldg + 2 * c2);
      if (NT) {
        t[i] = __builtin_nontemporal_load(pt);
        r[i] = __builtin_nontemporal_load(pr);
        g[i] = __builtin_nontemporal_load(pg);
      } else {
        t[i] = *pt;
        r[i] = *pr;
        g[i] = *pg;
      }
      m[i] = metric ? metric[d] : 1.0;
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    i64 d = d0 + i;
    if (d < D) {
      dvec2 rn, tn;
      double rx, ry;
      tn.x = kd_elem(t[i].x, r[i].x, g[i].x, m[i], metric != nullptr, eps, use_pre, pre, use_kick, kick, rx);
      tn.y = kd_elem(t[i].y, r[i].y, g[i].y, m[i], metric != nullptr, eps, use_pre, pre, use_kick, kick, ry);
      rn.x = rx;
      rn.y = ry;
      dvec2* qr = reinterpret_cast<dvec2*>(rho_out + d * ld + 2 * c2);
      dvec2* qt = reinterpret_cast<dvec2*>(th_out + d * ld + 2 * c2);
      if (NT) {
        __builtin_nontemporal_store(rn, qr);
        __builtin_nontemporal_store(tn, qt);
      } else {
        *qr = rn;
        *qt = tn;
      }
    }
  }
}

// any layout of the gradient, one chain per lane (odd C, unaligned views, exotic strides)
__global__ __launch_bounds__(KD_BLOCK) void k_kick_drift_s(
    const double* th_in, double* th_out, const double* rho_in, double* rho_out, i64 ld,
    const double* grad, i64 ldg_d, i64 ldg_c, const double* metric, double eps, int use_pre,
    double pre, int use_kick, double kick, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * KD_BLOCK + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * KD_ROWS;
  if (c >= C) return;
  double t[KD_ROWS], r[KD_ROWS], g[KD_ROWS], m[KD_ROWS];
#pragma unroll
  for (int i = 0; i < KD_ROWS; ++i) {
    i64 d = d0 + i;
    if (d < D) {
      t[i] = th_in[d * ld + c];
      r[i] = rho_in[d * ld + c];
      g[i] = grad[d * ldg_d + c * ldg_c];
      m[i] = metric ? metric[d] : 1.0;
    }
  }
#pragma unroll
  for (int i = 0; i < KD_ROWS; ++i) {
    i64 d = d0 + i;
    if (d < D) {
      double rn;
      double tn = kd_elem(t[i], r[i], g[i], m[i], metric != nullptr, eps, use_pre, pre, use_kick, kick, rn);
      rho_out[d * ld + c] = rn;
      th_out[d * ld + c] = tn;
    }
  }
}

// dimension-contiguous gradient (a row-major (C, D) model output): 64 x 64 tiles of the
// gradient are read coalesced along d, staged in LDS and re-read chain-major.  Row pitch
// 65 doubles keeps the transposed ds_read_b64 conflict-free within each 32-lane group.
constexpr int TR_TILE = 64;
__global__ __launch_bounds__(256) void k_kick_drift_tr(
    const double* th_in, double* th_out, const double* rho_in, double* rho_out, i64 ld,
    const double* grad, i64 ldg_c, const double* metric, double eps, int use_pre, double pre,
    int use_kick, double kick, i64 C, i64 D) {
  __shared__ double tile[TR_TILE][TR_TILE + 1];
  i64 c0 = (i64)blockIdx.x * TR_TILE, d0 = (i64)blockIdx.y * TR_TILE;
  int tx = threadIdx.x & 63, ty = bk_wave_id();
#pragma unroll 4
  for (int i = 0; i < TR_TILE / 4; ++i) {
    int cl = ty + 4 * i;
    i64 c = c0 + cl, d = d0 + tx;
    tile[cl][tx] = (c < C && d < D) ? grad[c * ldg_c + d] : 0.0;
  }
  __syncthreads();
  i64 c = c0 + tx;
  if (c >= C) return;
#pragma unroll 4
  for (int i = 0; i < TR_TILE / 4; ++i) {
    int dl = ty + 4 * i;
    i64 d = d0 + dl;
    if (d < D) {
      double rn;
      double m = metric ? metric[d] : 1.0;
      double tn = kd_elem(th_in[d * ld + c], rho_in[d * ld + c], tile[tx][dl], m, metric != nullptr, eps,
                          use_pre, pre, use_kick, kick, rn);
      rho_out[d * ld + c] = rn;
      th_out[d * ld + c] = tn;
    }
  }
}

// ---- first step of a delayed-rejection stage with gather of the active chains ---------
constexpr int PC_BLOCK = 64;  // per-chain kernels: one wavefront per workgroup
constexpr int RED_BLOCK_N = 4 * BK_WAVE;
constexpr int PC_UNROLL = 8;

__global__ __launch_bounds__(PC_BLOCK) void k_first_step_gather(
    const double* th_in, const double* rho_in, const double* g_in, i64 ld_in, const int32_t* idx,
    double* th_out, double* rho_out, i64 ld_out, const double* metric, double eps, double pre, i64 n,
    i64 D) {
  i64 j = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (j >= n) return;
  i64 src = idx ? (i64)idx[j] : j;
  for (i64 d0 = 0; d0 < D; d0 += PC_UNROLL) {
    double t[PC_UNROLL], r[PC_UNROLL], g[PC_UNROLL];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) {
        i64 o = (d0 + u) * ld_in + src;
        t[u] = th_in[o];
        r[u] = rho_in[o];
        g[u] = g_in[o];
      }
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u)
      if (d0 + u < D) {
        double rn;
        double m = metric ? metric[d0 + u] : 1.0;
        double tn = kd_elem(t[u], r[u], g[u], m, metric != nullptr, eps, 1, pre, 0, 0.0, rn);
        i64 o = (d0 + u) * ld_out + j;
        rho_out[o] = rn;
        th_out[o] = tn;
      }
  }
}

// ---- kick + drift over a lane set whose size may live on the device ---------------------
// The step of a delayed-rejection trajectory whose lane count only the device knows (and, with `idx`,
// the gathering first step): a workgroup serves 64 lanes (lane = chain of the compacted set), its four
// wavefronts and the gridDim.y row groups split the dimensions so that a wavefront has at most KN_U rows
// -- all of a step's loads in flight at once: these sets are a few per cent of the chains, the launch is
// one memory round trip deep.  The grid is sized for the host-side bound; workgroups past *n_dev exit.
constexpr int KN_U = 8;
__global__ __launch_bounds__(RED_BLOCK_N) void k_kick_drift_n(
    const double* th_in, const double* rho_in, const double* g_in, i64 ld_in, i64 ldg, const int32_t* idx,
    double* th_out, double* rho_out, i64 ld_out, const double* metric, double eps, int use_pre, double pre,
    int use_kick, double kick, i64 n_host, i64 D, const uint32_t* n_dev) {
  const i64 n = bk_lanes(n_host, n_dev);
  const i64 j0 = (i64)blockIdx.x * BK_WAVE;
  if (j0 >= n) return;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 j = j0 + lane;
  if (j >= n) return;
  const i64 src = idx ? (i64)idx[j] : j;
  const i64 stride = (i64)4 * gridDim.y;
  for (i64 d0 = (i64)blockIdx.y * 4 + w; d0 < D; d0 += stride * KN_U) {
    double t[KN_U], r[KN_U], g[KN_U], m[KN_U];
#pragma unroll
    for (int u = 0; u < KN_U; ++u) {
      const i64 d = d0 + u * stride;
      if (d < D) {
        t[u] = th_in[d * ld_in + src];
        r[u] = rho_in[d * ld_in + src];
        g[u] = g_in[d * ldg + src];
        m[u] = metric ? metric[d] : 1.0;
      }
    }
#pragma unroll
    for (int u = 0; u < KN_U; ++u) {
      const i64 d = d0 + u * stride;
      if (d < D) {
        double rn;
        double tn = kd_elem(t[u], r[u], g[u], m[u], metric != nullptr, eps, use_pre, pre, use_kick, kick, rn);
        rho_out[d * ld_out + j] = rn;
        th_out[d * ld_out + j] = tn;
      }
    }
  }
}

// ---- final half-kick + kinetic energy ---------------------------------------------------
// Per-chain reductions: a workgroup of RED_WAVES wavefronts serves 64 chains (lane = chain);
// wavefront w owns the contiguous quarter of the dimensions [w*Dq, (w+1)*Dq) and sums it
// sequentially, the quarters are combined through LDS in the fixed order ((p0+p1)+p2)+p3.
// The order depends on D only, never on the number of chains or the grid.
constexpr int RED_WAVES = 4;
constexpr int RED_BLOCK = RED_WAVES * BK_WAVE;
constexpr int FIN_UNROLL = 8;

__device__ __forceinline__ double red_combine(double (&part)[RED_WAVES][BK_WAVE], int w, int lane, double p) {
  part[w][lane] = p;
  __syncthreads();
  double s = part[0][lane];
#pragma unroll
  for (int k = 1; k < RED_WAVES; ++k) s = s + part[k][lane];
  return s;
}

// n_dev / level: the trajectory of a delayed-rejection level whose lane count lives on the device -- the
// launch is sized for C (a bound), works on min(C, *n_dev) lanes, and sets the level up for accept()
// (bk_dr_level_begin: H = joint(logp, kin), h = 0, live = 1) and counts its lanes in the same pass.
struct FinishLevel {
  const double* logp;
  double* H;
  double* h;
  uint8_t* live;
  uint32_t* lanes_out;
  unsigned long long* lanes_total;
};

__global__ __launch_bounds__(RED_BLOCK) void k_finish(const double* rho_in, double* rho_out, i64 ld,
                                                      const double* grad, i64 ldg_d, i64 ldg_c,
                                                      const double* metric, double half, int negate,
                                                      double* kin_out, i64 C_host, i64 D, const uint32_t* n_dev,
                                                      FinishLevel lv) {
  __shared__ double part[RED_WAVES][BK_WAVE];
  constexpr int PC_UNROLL = FIN_UNROLL;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 C = bk_lanes(C_host, n_dev);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (lv.lanes_out) *lv.lanes_out = (uint32_t)C;
    if (lv.lanes_total) *lv.lanes_total += (unsigned long long)C;  // one writer per launch, launches are stream-ordered
  }
  if ((i64)blockIdx.x * BK_WAVE >= C) return;  // (whole workgroup past the set: uniform)
  const i64 c = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 Dq = (D + RED_WAVES - 1) / RED_WAVES;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  double kin = 0.0;
  if (c < C) {
    for (i64 d0 = dlo; d0 < dhi; d0 += PC_UNROLL) {
      double r[PC_UNROLL], g[PC_UNROLL];
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < dhi) {
          r[u] = rho_in[(d0 + u) * ld + c];
          g[u] = grad ? grad[(d0 + u) * ldg_d + c * ldg_c] : 0.0;
        }
#pragma unroll
      for (int u = 0; u < PC_UNROLL; ++u)
        if (d0 + u < dhi) {
          double m = metric ? metric[d0 + u] : 1.0;
          double t = metric ? m * g[u] : g[u];
          double v = grad ? r[u] + half * t : r[u];  // grad NULL: kinetic energy of rho_in as is
          if (negate) v = -v;
          if (rho_out) rho_out[(d0 + u) * ld + c] = v;
          double mv = metric ? m * v : v;
          kin = kin + v * mv;
        }
    }
  }
  if (!kin_out) return;  // uniform: no barrier needed without the reduction
  double s = red_combine(part, w, lane, kin);
  if (w == 0 && c < C) {
    const double k = 0.5 * s;
    kin_out[c] = k;
    if (lv.H) {
      lv.H[c] = dr_joint(lv.logp[c], k);
      lv.h[c] = 0.0;
      lv.live[c] = 1;
    }
  }
}

// The same with two chains (16 B) per lane: 128 chains per workgroup, 1 KiB per wavefront and row instead
// of 512 B (8-byte-per-lane streams reach 0.5-0.7x the rate of 16-byte ones on gfx950: 344 us for the
// 1.07 GB of a config-3 launch).  Each component runs the scalar kernel's operation sequence: same values.
__global__ __launch_bounds__(RED_BLOCK) void k_finish_v2(const double* rho_in, double* rho_out, i64 ld,
                                                         const double* grad, i64 ldg, const double* metric,
                                                         double half, int negate, double* kin_out, i64 C2, i64 D) {
  __shared__ dvec2 part[RED_WAVES][BK_WAVE];
  constexpr int U = FIN_UNROLL;
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 c2 = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 Dq = (D + RED_WAVES - 1) / RED_WAVES;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  dvec2 kin = {0.0, 0.0};
  if (c2 < C2) {
    for (i64 d0 = dlo; d0 < dhi; d0 += U) {
      dvec2 r[U], g[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) {
          r[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(rho_in + (d0 + u) * ld + 2 * c2));
          if (grad) g[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(grad + (d0 + u) * ldg + 2 * c2));
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) {
          const double m = metric ? metric[d0 + u] : 1.0;
          dvec2 v;
          {
            double t = metric ? m * g[u].x : g[u].x;
            v.x = grad ? r[u].x + half * t : r[u].x;
            t = metric ? m * g[u].y : g[u].y;
            v.y = grad ? r[u].y + half * t : r[u].y;
          }
          if (negate) {
            v.x = -v.x;
            v.y = -v.y;
          }
          if (rho_out) *reinterpret_cast<dvec2*>(rho_out + (d0 + u) * ld + 2 * c2) = v;
          const double mx = metric ? m * v.x : v.x, my = metric ? m * v.y : v.y;
          kin.x = kin.x + v.x * mx;
          kin.y = kin.y + v.y * my;
        }
    }
  }
  if (!kin_out) return;  // uniform: no barrier needed without the reduction
  part[w][lane] = kin;
  __syncthreads();
  if (w == 0 && c2 < C2) {
    dvec2 s = part[0][lane];
#pragma unroll
    for (int k = 1; k < RED_WAVES; ++k) {
      s.x = s.x + part[k][lane].x;
      s.y = s.y + part[k][lane].y;
    }
    s.x = 0.5 * s.x;
    s.y = 0.5 * s.y;
    *reinterpret_cast<dvec2*>(kin_out + 2 * c2) = s;
  }
}

// ---- accept ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mh_accept(int mode, double* lp_cur, const double* a_cur,
                                                   const double* lp_prop, const double* a_prop,
                                                   const double* log_u, uint8_t* mask, double* ret,
                                                   uint32_t* count, i64 C) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  bool acc = false;
  if (c < C) {
    double l0 = lp_cur[c], l1 = lp_prop[c];
    double a0 = a_cur ? a_cur[c] : 0.0, a1 = a_prop ? a_prop[c] : 0.0;
    double r0, r1;
    if (mode == BK_ACCEPT_HMC) {
      double h0 = l0 - a0, h1 = l1 - a1;  // hmc.py:36-38
      acc = log_u[c] < h1 - h0;           // hmc.py:60
      r0 = h0;
      r1 = h1;
    } else {
      acc = log_u[c] < (l1 - l0) + (a1 - a0);  // metropolis.py:70-76
      r0 = l0;
      r1 = l1;
    }
    if (mask) mask[c] = acc ? 1 : 0;
    if (ret) ret[c] = acc ? r1 : r0;
    if (acc) lp_cur[c] = l1;
  }
  if (count) {
    // wavefront ballot -> one atomic per 64 chains
    unsigned long long b = __ballot(acc);
    if ((threadIdx.x & (BK_WAVE - 1)) == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
  }
}

// ---- masked column copy -----------------------------------------------------------------
constexpr int SEL_ROWS = 2;
// copy0 (optional): the selected array as it stands after the select, for every chain -- the
// stable array sample() hands back -- written in the same pass instead of by a second copy.
__global__ __launch_bounds__(256) void k_select(const uint8_t* mask, double* dst0, const double* src0,
                                                double* dst1, const double* src1, double* copy0, i64 ld,
                                                i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * SEL_ROWS;
  if (c >= C) return;
  const bool m = mask[c] != 0;
  if (!m && !copy0) return;
  double a[SEL_ROWS], b[SEL_ROWS];
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) {
      a[i] = m ? src0[(d0 + i) * ld + c] : dst0[(d0 + i) * ld + c];
      if (dst1 && m) b[i] = src1[(d0 + i) * ld + c];
    }
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) {
      if (m) dst0[(d0 + i) * ld + c] = a[i];
      if (dst1 && m) dst1[(d0 + i) * ld + c] = b[i];
      if (copy0) copy0[(d0 + i) * ld + c] = a[i];
    }
}

// Two chains (16 B) per lane, written as a BLEND: every lane pair re-writes its 16 bytes, rejected
// chains with their own old values.  Skipping the rejected chains instead (masked or 8-byte stores)
// leaves holes inside HBM sectors, and each holed sector costs a read-modify-write in the memory
// system: measured 881 -> 623 us for two array pairs + output copy at 65,536 x 1024, accept 0.8
// (tools/select_bench.py); the blend runs at the streaming rate (6.0 TB/s) at any accept rate.
template <bool COPY>
__global__ __launch_bounds__(256) void k_select_v2(const uint8_t* mask, double* dst0, const double* src0,
                                                   double* dst1, const double* src1, double* copy0, i64 ld,
                                                   i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * SEL_ROWS;
  if (c2 >= C2) return;
  const bool m0 = mask[2 * c2] != 0, m1 = mask[2 * c2 + 1] != 0;
  const bool any = m0 || m1, both = m0 && m1;
  dvec2 a[SEL_ROWS], b[SEL_ROWS];
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) {
      const i64 o = (d0 + i) * ld + 2 * c2;
      if (any) {
        a[i] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(src0 + o));
        if (dst1) b[i] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(src1 + o));
      }
      if (!both) {
        dvec2 old = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(dst0 + o));
        if (!m0) a[i].x = old.x;
        if (!m1) a[i].y = old.y;
        if (dst1) {
          old = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(dst1 + o));
          if (!m0) b[i].x = old.x;
          if (!m1) b[i].y = old.y;
        }
      }
    }
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) {
      const i64 o = (d0 + i) * ld + 2 * c2;
      __builtin_nontemporal_store(a[i], reinterpret_cast<dvec2*>(dst0 + o));
      if (dst1) __builtin_nontemporal_store(b[i], reinterpret_cast<dvec2*>(dst1 + o));
      if (COPY) __builtin_nontemporal_store(a[i], reinterpret_cast<dvec2*>(copy0 + o));
    }
}

// out = mask ? b : a, nothing else written: the select for samplers that REBIND their state array every
// draw (the reference's `self._theta = theta_prop`, hmc.py:61) -- `out` becomes the state and is what
// sample() returns, so the old state is only read: ~9 B read + 8 B written per element instead of
// 9 + 16 for the in-place select with its returned copy.
__global__ __launch_bounds__(256) void k_blend(const uint8_t* mask, const double* a, const double* b, double* out,
                                               i64 ld, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * SEL_ROWS;
  if (c >= C) return;
  const double* src = mask[c] ? b : a;
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) out[(d0 + i) * ld + c] = src[(d0 + i) * ld + c];
}

__global__ __launch_bounds__(256) void k_blend_v2(const uint8_t* mask, const double* a, const double* b,
                                                  double* out, i64 ld, i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * SEL_ROWS;
  if (c2 >= C2) return;
  const bool m0 = mask[2 * c2] != 0, m1 = mask[2 * c2 + 1] != 0;
  dvec2 v[SEL_ROWS];
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D) {
      const i64 o = (d0 + i) * ld + 2 * c2;
      if (m0 || m1) v[i] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(b + o));
      if (!(m0 && m1)) {
        dvec2 old = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(a + o));
        if (!m0) v[i].x = old.x;
        if (!m1) v[i].y = old.y;
      }
    }
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (d0 + i < D)
      __builtin_nontemporal_store(v[i], reinterpret_cast<dvec2*>(out + (d0 + i) * ld + 2 * c2));
}

// ---- MALA proposal log densities --------------------------------------------------------
__global__ __launch_bounds__(RED_BLOCK) void k_mala_logq(const double* th, const double* g,
                                                         const double* thp, const double* gp, i64 ld,
                                                         double eps, double* fwd, double* rev, i64 C,
                                                         i64 D) {
  __shared__ double part_f[RED_WAVES][BK_WAVE];
  __shared__ double part_r[RED_WAVES][BK_WAVE];
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 c = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 Dq = (D + RED_WAVES - 1) / RED_WAVES;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  double sf = 0.0, sr = 0.0;
  constexpr int U = 4;
  if (c < C) {
    for (i64 d0 = dlo; d0 < dhi; d0 += U) {
      double a[U], b[U], p[U], q[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) {
          i64 o = (d0 + u) * ld + c;
          a[u] = th[o];
          b[u] = g[o];
          p[u] = thp[o];
          q[u] = gp[o];
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d0 + u < dhi) {
          double xf = (p[u] - a[u]) - eps * b[u];  // mala.py:78
          double xr = (a[u] - p[u]) - eps * q[u];
          sf = sf + xf * xf;
          sr = sr + xr * xr;
        }
    }
  }
  double tf = red_combine(part_f, w, lane, sf);
  double tr = red_combine(part_r, w, lane, sr);
  if (w == 0 && c < C) {
    double k = -0.25 / eps;  // mala.py:79
    fwd[c] = k * tf;
    rev[c] = k * tr;
  }
}

// ---- layout change through LDS tiles ------------------------------------------------------
__global__ __launch_bounds__(256) void k_relayout(const double* src, i64 s_d, i64 s_c, double* dst,
                                                  i64 t_d, i64 t_c, i64 C, i64 D) {
  __shared__ double tile[TR_TILE][TR_TILE + 1];
  i64 c0 = (i64)blockIdx.x * TR_TILE, d0 = (i64)blockIdx.y * TR_TILE;
  int tx = threadIdx.x & 63, ty = bk_wave_id();
  // read with the lane index along whichever source axis is contiguous
  bool src_d_fast = (s_d == 1);
#pragma unroll 4
  for (int i = 0; i < TR_TILE / 4; ++i) {
    int slow = ty + 4 * i;
    int cl = src_d_fast ? slow : tx, dl = src_d_fast ? tx : slow;
    i64 c = c0 + cl, d = d0 + dl;
    if (c < C && d < D) tile[cl][dl] = src[d * s_d + c * s_c];
  }
  __syncthreads();
  bool dst_d_fast = (t_d == 1);
#pragma unroll 4
  for (int i = 0; i < TR_TILE / 4; ++i) {
    int slow = ty + 4 * i;
    int cl = dst_d_fast ? slow : tx, dl = dst_d_fast ? tx : slow;
    i64 c = c0 + cl, d = d0 + dl;
    if (c < C && d < D) dst[d * t_d + c * t_c] = tile[cl][dl];
  }
}

}  // namespace

extern "C" {

int bk_leapfrog_kick_drift(const double* theta_in, double* theta_out, const double* rho_in,
                           double* rho_out, int64_t ld, const double* grad, int64_t ldg_d,
                           int64_t ldg_c, const double* metric, double eps, int use_pre, double pre,
                           int use_kick, double kick, int64_t C, int64_t D, void* stream) {
  if (!theta_in || !theta_out || !rho_in || !rho_out || !grad || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (ldg_c == 1) {
    bool vec = (C % 2 == 0) && (ld % 2 == 0) && (ldg_d % 2 == 0) && bk_aligned16(theta_in) &&
               bk_aligned16(theta_out) && bk_aligned16(rho_in) && bk_aligned16(rho_out) &&
               bk_aligned16(grad);
    if (vec) {
      // Streams much larger than the 256 MiB Infinity Cache use non-temporal loads/stores
      // (measured +8 % on MI355X: 6.49 vs 6.0 TB/s); cache-resident tiles use plain
      // accesses (7.1 TB/s out of the Infinity Cache).  BK_KD_VARIANT overrides (tuning).
      static const int forced = []() { const char* e = getenv("BK_KD_VARIANT"); return e ? atoi(e) : -1; }();
      int v = forced >= 0 ? forced : (bk_streams_past_llc(5 * C * D) ? 4 : 5);
#define BK_KD_LAUNCH(ROWS, NT)                                                                          \
  do {                                                                                                  \
    dim3 grid((unsigned)bk_cdiv(C / 2, KD_BLOCK), (unsigned)bk_cdiv(D, ROWS));                          \
    k_kick_drift_v2<ROWS, NT><<<grid, dim3(KD_BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, \
                                                              grad, ldg_d, metric, eps, use_pre, pre,   \
                                                              use_kick, kick, C / 2, D);                \
  } while (0)
      switch (v) {
        case 1: BK_KD_LAUNCH(2, false); break;
        case 2: BK_KD_LAUNCH(8, false); break;
        case 3: BK_KD_LAUNCH(4, true); break;
        case 4: BK_KD_LAUNCH(2, true); break;
        case 5: BK_KD_LAUNCH(1, false); break;
        case 6: BK_KD_LAUNCH(1, true); break;
        default: BK_KD_LAUNCH(4, false); break;
      }
#undef BK_KD_LAUNCH
      BK_RETURN_LAUNCH_STATUS();
    }
  } else if (ldg_d == 1 && C >= TR_TILE / 2 && D >= TR_TILE / 2) {
    dim3 grid((unsigned)bk_cdiv(C, TR_TILE), (unsigned)bk_cdiv(D, TR_TILE));
    k_kick_drift_tr<<<grid, dim3(256), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, grad, ldg_c, metric,
                                               eps, use_pre, pre, use_kick, kick, C, D);
    BK_RETURN_LAUNCH_STATUS();
  }
  dim3 grid((unsigned)bk_cdiv(C, KD_BLOCK), (unsigned)bk_cdiv(D, KD_ROWS));
  k_kick_drift_s<<<grid, dim3(KD_BLOCK), 0, s>>>(theta_in, theta_out, rho_in, rho_out, ld, grad, ldg_d, ldg_c,
                                                 metric, eps, use_pre, pre, use_kick, kick, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

static unsigned kn_row_groups(i64 D) {
  // at most KN_U rows per wavefront (4 wavefronts per workgroup, gridDim.y row groups)
  i64 g = bk_cdiv(D, 4 * KN_U);
  return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

int bk_leapfrog_kick_drift_n(const double* theta_in, double* theta_out, const double* rho_in, double* rho_out,
                             int64_t ld, const double* grad, int64_t ldg_d, int64_t ldg_c, const double* metric,
                             double eps, int use_pre, double pre, int use_kick, double kick, int64_t C, int64_t D,
                             const uint32_t* n_dev, void* stream) {
  if (!n_dev)
    return bk_leapfrog_kick_drift(theta_in, theta_out, rho_in, rho_out, ld, grad, ldg_d, ldg_c, metric, eps, use_pre,
                                  pre, use_kick, kick, C, D, stream);
  if (!theta_in || !theta_out || !rho_in || !rho_out || !grad || C < 0 || D < 0) return BK_E_ARG;
  if (ldg_c != 1 && C > 1) return BK_E_ARG;  // (a counted lane set lives in the library's own chain-contiguous buffers)
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(C, BK_WAVE), kn_row_groups(D));
  k_kick_drift_n<<<grid, dim3(RED_BLOCK_N), 0, bk_stream(stream)>>>(theta_in, rho_in, grad, ld, ldg_d, nullptr, theta_out,
                                                                    rho_out, ld, metric, eps, use_pre, pre, use_kick,
                                                                    kick, C, D, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_leapfrog_first_step_gather(const double* theta_in, const double* rho_in, const double* grad_in,
                                    int64_t ld_in, const int32_t* src_index, double* theta_out, double* rho_out,
                                    int64_t ld_out, const double* metric, double eps, double pre, int64_t n,
                                    int64_t D, const uint32_t* n_dev, void* stream) {
  if (!theta_in || !rho_in || !grad_in || !theta_out || !rho_out || n < 0 || D < 0) return BK_E_ARG;
  if (ld_out < n) return BK_E_ALIGN;
  if (n == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(n, BK_WAVE), kn_row_groups(D));
  k_kick_drift_n<<<grid, dim3(RED_BLOCK_N), 0, bk_stream(stream)>>>(theta_in, rho_in, grad_in, ld_in, ld_in, src_index,
                                                                    theta_out, rho_out, ld_out, metric, eps, 1, pre, 0,
                                                                    0.0, n, D, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_leapfrog_finish_level(const double* rho_in, double* rho_out, int64_t ld, const double* grad, int64_t ldg_d,
                             int64_t ldg_c, const double* metric, double half, int negate, double* kin_out,
                             int64_t C, int64_t D, const uint32_t* n_dev, const double* logp, double* H_out,
                             double* h_out, uint8_t* live_out, uint32_t* lanes_out, uint64_t* lanes_total,
                             void* stream) {
  if (!rho_in || C < 0 || D < 0) return BK_E_ARG;
  if (H_out && (!logp || !h_out || !live_out || !kin_out)) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  hipStream_t s = bk_stream(stream);
  if (C == 0) {
    if (lanes_out) return (int)hipMemsetAsync(lanes_out, 0, sizeof(uint32_t), s);
    return BK_OK;
  }
  FinishLevel lv = {logp, H_out, h_out, live_out, lanes_out, reinterpret_cast<unsigned long long*>(lanes_total)};
  const bool plain = !n_dev && !H_out && !lanes_out && !lanes_total;
  // (two chains per lane halves the number of workgroups: only for arrays that do not live in L2 -- at
  // 32,768 x 101 the one-chain form is faster, 7.4 vs 9.1 us)
  const bool vec = plain && C % 2 == 0 && ld % 2 == 0 && C * D >= ((i64)1 << 22) && bk_aligned16(rho_in) &&
                   (!rho_out || bk_aligned16(rho_out)) && (!kin_out || bk_aligned16(kin_out)) &&
                   (!grad || (ldg_c == 1 && ldg_d % 2 == 0 && bk_aligned16(grad)));
  if (vec)
    k_finish_v2<<<dim3((unsigned)bk_cdiv(C / 2, BK_WAVE)), dim3(RED_BLOCK), 0, s>>>(
        rho_in, rho_out, ld, grad, ldg_d, metric, half, negate, kin_out, C / 2, D);
  else
    k_finish<<<dim3((unsigned)bk_cdiv(C, BK_WAVE)), dim3(RED_BLOCK), 0, s>>>(
        rho_in, rho_out, ld, grad, ldg_d, ldg_c, metric, half, negate, kin_out, C, D, n_dev, lv);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_leapfrog_finish(const double* rho_in, double* rho_out, int64_t ld, const double* grad,
                       int64_t ldg_d, int64_t ldg_c, const double* metric, double half, int negate,
                       double* kin_out, int64_t C, int64_t D, void* stream) {
  return bk_leapfrog_finish_level(rho_in, rho_out, ld, grad, ldg_d, ldg_c, metric, half, negate, kin_out, C, D,
                                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

int bk_mh_accept(int mode, double* lp_cur, const double* a_cur, const double* lp_prop,
                 const double* a_prop, const double* log_u, uint8_t* accept_mask, double* ret,
                 uint32_t* accept_count, int64_t C, void* stream) {
  if (!lp_cur || !lp_prop || !log_u || C < 0) return BK_E_ARG;
  if (mode != BK_ACCEPT_HMC && mode != BK_ACCEPT_MALA) return BK_E_ARG;
  if (C == 0) return BK_OK;
  k_mh_accept<<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, bk_stream(stream)>>>(
      mode, lp_cur, a_cur, lp_prop, a_prop, log_u, accept_mask, ret, accept_count, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_select_columns(const uint8_t* mask, double* dst0, const double* src0, double* dst1,
                      const double* src1, double* copy0, int64_t ld, int64_t C, int64_t D, void* stream) {
  if (!mask || !dst0 || !src0 || (dst1 && !src1) || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(dst0) && bk_aligned16(src0) &&
      (!dst1 || (bk_aligned16(dst1) && bk_aligned16(src1))) && (!copy0 || bk_aligned16(copy0))) {
    dim3 grid((unsigned)bk_cdiv(C / 2, 256), (unsigned)bk_cdiv(D, SEL_ROWS));
    if (copy0)
      k_select_v2<true><<<grid, dim3(256), 0, s>>>(mask, dst0, src0, dst1, src1, copy0, ld, C / 2, D);
    else
      k_select_v2<false><<<grid, dim3(256), 0, s>>>(mask, dst0, src0, dst1, src1, nullptr, ld, C / 2, D);
  } else {
    dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, SEL_ROWS));
    k_select<<<grid, dim3(256), 0, s>>>(mask, dst0, src0, dst1, src1, copy0, ld, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_blend_columns(const uint8_t* mask, const double* a, const double* b, double* out, int64_t ld, int64_t C,
                     int64_t D, void* stream) {
  if (!mask || !a || !b || !out || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  if (C % 2 == 0 && ld % 2 == 0 && bk_aligned16(a) && bk_aligned16(b) && bk_aligned16(out)) {
    dim3 grid((unsigned)bk_cdiv(C / 2, 256), (unsigned)bk_cdiv(D, SEL_ROWS));
    k_blend_v2<<<grid, dim3(256), 0, s>>>(mask, a, b, out, ld, C / 2, D);
  } else {
    dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, SEL_ROWS));
    k_blend<<<grid, dim3(256), 0, s>>>(mask, a, b, out, ld, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_mala_logq(const double* theta, const double* grad, const double* theta_prop,
                 const double* grad_prop, int64_t ld, double eps, double* lp_forward,
                 double* lp_reverse, int64_t C, int64_t D, void* stream) {
  if (!theta || !grad || !theta_prop || !grad_prop || !lp_forward || !lp_reverse || C < 0 || D < 0)
    return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_mala_logq<<<dim3((unsigned)bk_cdiv(C, BK_WAVE)), dim3(RED_BLOCK), 0, bk_stream(stream)>>>(
      theta, grad, theta_prop, grad_prop, ld, eps, lp_forward, lp_reverse, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_relayout(const double* src, int64_t lds_d, int64_t lds_c, double* dst, int64_t ldd_d,
                int64_t ldd_c, int64_t C, int64_t D, void* stream) {
  if (!src || !dst || C < 0 || D < 0) return BK_E_ARG;
  if (C == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(C, TR_TILE), (unsigned)bk_cdiv(D, TR_TILE));
  k_relayout<<<grid, dim3(256), 0, bk_stream(stream)>>>(src, lds_d, lds_c, dst, ldd_d, ldd_c, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
