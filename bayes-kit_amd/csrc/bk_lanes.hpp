// Lane-spread densities: the library's trajectory and gradient kernels for ANY log density of the shape
//
//     log p(theta) = F( head coordinates, sums over the remaining ("spread") coordinates of per-row terms )
//
// (hierarchical models: a few shared hyper-parameters + many exchangeable rows; Neal's funnel is HEAD = 1).
// The density is the ONLY model-specific code: a struct with
//
//     static constexpr int HEAD;                       // leading coordinates every lane of a chain holds
//     template <class L> static double eval(L& c, const double* params);   // returns log p
//
// written against a lane context `c`:
//
//     c.dims()                  D
//     c.head(i)                 theta_i, i < HEAD (i a constant)
//     c.sum(f)                  sum over the spread rows d >= HEAD of f(theta_d, d), in the library's canonical
//                               order (below); the same value in every lane of the chain
//     c.grad_head(i, g)         d log p / d theta_i
//     c.wants_logp()            (optional) false where the returned value is discarded: sums that only feed it may be skipped
//     c.grad(f)                 d log p / d theta_d = f(theta_d, d) for every spread row; call it ONCE and LAST
//                               (a trajectory context advances the row as its gradient is delivered; a sum() after
//                               grad() traps)
//
// The integrator text is the library's: bayes_kit/drghmc.py:253-289 (leapfrog), :319-346 (proposal map),
// :391-446 (accept recursion, first ghost) are kernels of THIS header, instantiated once for the built-in funnel
// (bk_targets.hip) and once per CTarget.from_source(form="lanes") density (the generated translation unit includes
// this file; the user's function is the only inlined callee).
//
// Canonical summation order (results do not depend on how many chains are in flight, on the grid, or on which
// geometry runs).  Spread row d belongs to class c = (d - HEAD) mod 16, slot i = (d - HEAD) div 16:
//     cs[c] = sum over i, in order, of f(theta[HEAD + c + 16 i])       (16 class sums)
//     q[g]  = ((cs[g] + cs[g+4]) + cs[g+8]) + cs[g+12],  g = 0..3      (4 group sums)
//     s     = ((q[0] + q[1]) + q[2]) + q[3]
// LPC adjacent lanes of ONE wavefront serve a chain and s is reduced with DPP moves inside the wavefront -- no
// LDS, no workgroup barrier in the leapfrog loop:
//     LPC = 4  (throughput: every chain): 16 chains per wavefront; lane p of a chain's quad holds the classes of
//              group p (p, p+4, p+8, p+12: up to 32 rows), forms q[p] itself, s over the quad.
//     LPC = 8  : lane p holds classes p and p + 8.
//     LPC = 16 (latency: sparse, long later stages): 4 chains per wavefront, one 16-lane DPP row each; lane p holds
//              class p (up to 8 rows).
// Every lane integrates the head coordinates redundantly; one lane per chain writes them.
#pragma once
#include "bk_common.hpp"
#include <math.h>
#include <stdlib.h>

namespace bkl {

constexpr int WAVES = 4;
constexpr int CLASSES = 16;
constexpr int MAX_SLOTS = 8;  // slots per class held in registers: D - HEAD <= 128
constexpr int MAX_ROWS = CLASSES * MAX_SLOTS;
constexpr int BLOCK = WAVES * BK_WAVE;
constexpr int MAX_HEAD = 8;

// ---- DPP moves of a double (two 32-bit halves) ------------------------------------------------------------------
// Lanes the control does not reach (row / bank masks, a shift whose source lies outside the 16-lane row) keep `old`.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHL = 0x100;  // + n: lane i reads lane i + n of its row
constexpr int DPP_ROW_SHR = 0x110;  // + n: lane i reads lane i - n of its row
constexpr int DPP_ROW_NEWBCAST = 0x150;  // + n: every lane reads lane n of its row (64-bit operands allowed)
// the same for controls under which every lane that matters has a source lane: no `old` operand, so no copy
// of the source in front of the move (lanes without a source read 0)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_all(double src) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(src), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(src), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ double quad_bcast(double x) {  // lane K of every quad, to the whole quad
  return dpp_f64_all<K * 0x55>(x);
}

//   LPC = 4 : p = lane % 4 is the class GROUP; register slot u = k*SL + i holds class p + 4k, slot i.
//   LPC = 8 : p = lane % 8;                    register slot u = k*SL + i holds class p + 8k, slot i (k = 0, 1).
//   LPC = 16: p = lane % 16 is the CLASS;      register slot u = i     holds class p,      slot i.
// A lane's rows are HEAD + p + off(u) with a lane-independent off(u): row addresses are a per-lane 32-bit
// offset (row HEAD + p of the lane's chain) on top of a wavefront-uniform row base, and only the LAST slot
// of a class can run past D (SL is exactly ceil((D-HEAD)/16)): every other row needs no guard.
template <int LPC, int SL>
struct Geo {
  static_assert(LPC == 4 || LPC == 8 || LPC == 16, "a chain is served by a quad, half a DPP row or a DPP row");
  static constexpr int KC = CLASSES / LPC;      // classes per lane
  static constexpr int NU = KC * SL;            // register slots per lane
  static constexpr int CHAINS = BK_WAVE / LPC;  // chains per wavefront
  __host__ __device__ static constexpr int off(int u) {  // row of slot u, relative to the lane's first row
    return LPC * (u / SL) + CLASSES * (u % SL);  // (LPC = 16: one class per lane, u / SL = 0)
  }
  __host__ __device__ static constexpr bool last_slot(int u) { return u % SL == SL - 1; }
};

// s (canonical order) of the chain this lane serves, from the lane's class sums; valid in EVERY lane
// of the chain.  All 64 lanes must be active.
template <int LPC>
__device__ __forceinline__ double reduce_lanes(const double* cs) {
  double q;
  if (LPC == 4) {
    q = ((cs[0] + cs[1]) + cs[2]) + cs[3];  // this lane holds classes p, p+4, p+8, p+12
  } else if (LPC == 8) {
    // lane p of the 8-lane group holds cs[p] and cs[p+8]; lanes 0..3 fetch cs[p+4], cs[p+12] from lane p + 4
    const double b = dpp_f64_all<DPP_ROW_SHL + 4>(cs[0]);
    const double d = dpp_f64_all<DPP_ROW_SHL + 4>(cs[1]);
    q = ((cs[0] + b) + cs[1]) + d;
  } else {
    // lane p of the row holds cs[p]; in lanes 0..3: q[p] = ((cs[p] + cs[p+4]) + cs[p+8]) + cs[p+12]
    // (the other twelve lanes compute something nobody reads)
    const double b = dpp_f64_all<DPP_ROW_SHL + 4>(cs[0]);
    const double c = dpp_f64_all<DPP_ROW_SHL + 8>(cs[0]);
    const double d = dpp_f64_all<DPP_ROW_SHL + 12>(cs[0]);
    q = ((cs[0] + b) + c) + d;
    // lane 0 of the row adds q[0..3] in order and hands s to the whole row in ONE 64-bit move (row_newbcast): 19
    // instructions for the reduction instead of the 25 of four quad broadcasts + two masked shifts -- the latency geometry
    // runs one wavefront per SIMD, where the leapfrog step costs what it issues
    const double s0 = ((q + dpp_f64_all<DPP_ROW_SHL + 1>(q)) + dpp_f64_all<DPP_ROW_SHL + 2>(q)) + dpp_f64_all<DPP_ROW_SHL + 3>(q);
    return __builtin_amdgcn_update_dpp(s0, s0, DPP_ROW_NEWBCAST + 0, 0xF, 0xF, false);
  }
  // lane p of the quad holds q[p]
  double s = ((quad_bcast<0>(q) + quad_bcast<1>(q)) + quad_bcast<2>(q)) + quad_bcast<3>(q);
  if (LPC == 8) {
    // s is right in lanes 0..3 of each 8-lane group: hand it to lanes 4..7 (banks 1 and 3 of the row)
    s = dpp_f64<DPP_ROW_SHR + 4, 0xF, 0xA>(s, s);
  }
  return s;
}

// class sums of term(u) over this lane's rows (tail_ok[k]: the last slot of class k exists), each class
// sequential in its slots
template <class G, int SL, class T>
__device__ __forceinline__ void class_sums(double* cs, const bool* tail_ok, T&& term) {
#pragma unroll
  for (int k = 0; k < G::KC; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int u = k * SL + i;
      if (!G::last_slot(u) || tail_ok[k]) acc = acc + term(u);
    }
    cs[k] = acc;
  }
}

// ---- the lane context of a trajectory: rows in registers, the gradient delivered INTO the kick -----------------
//   r  += hk * (metric * grad)            (drghmc.py:277 / :281 / :286)
//   x  += hd * r     when DRIFT           (drghmc.py:278 / :282)
//   g_out <- grad    when STORE           (the proposal's end point keeps its gradient, drghmc.py:285)
// MTR: the 16-lane geometry keeps its rows' metric entries in registers (<= 8 rows), the others re-read them (L1).
template <int LPC, int SL, bool HM, int HEAD, bool STORE, bool DRIFT>
struct TrajCtx {
  using G = Geo<LPC, SL>;
  static constexpr int NU = G::NU, KC = G::KC;
  static constexpr bool MTR = HM && LPC == 16;
  double* x;             // [NU] this lane's rows
  double* r;             // [NU] their momenta
  const double* v;       // [HEAD] head coordinates
  double* rv;            // [HEAD] their momenta
  double* gv;            // [HEAD] out: the head gradient
  const bool* tail_ok;   // [KC]
  const double* mt;      // [NU] row metric entries (MTR)
  const double* mvh;     // [HEAD] head metric entries
  const double* metric;  // device array (or NULL)
  double hk, hd;
  int pos;
  i64 D;
  double* g_out;  // STORE
  i64 ld_out;
  uint32_t bo_out;
  bool on;
  bool rows_done;

  __device__ __forceinline__ i64 dims() const { return D; }
  __device__ __forceinline__ double head(int i) const { return v[i]; }
  // whether the returned log density is used (only at a trajectory's end point): a density whose gradient needs none of the
  // sums that make up its value may skip them otherwise
  __device__ __forceinline__ bool wants_logp() const { return STORE || !DRIFT; }
  __device__ __forceinline__ i64 row(int u) const { return (i64)(HEAD + pos + G::off(u)); }
  __device__ __forceinline__ bool ok(int u) const { return !G::last_slot(u) || tail_ok[u / SL]; }
  __device__ __forceinline__ double metric_of(int u) const { return MTR ? mt[MTR ? u : 0] : metric[HEAD + pos + G::off(u)]; }

  template <class F>
  __device__ __forceinline__ double sum(F&& f) {
    if (rows_done) __builtin_trap();  // (folds away: the flag is a compile-time constant after inlining)
    double cs[KC];
    class_sums<G, SL>(cs, tail_ok, [&](int u) { return f(x[u], row(u)); });
    return reduce_lanes<LPC>(cs);
  }
  __device__ __forceinline__ void grad_head(int i, double g) {
    gv[i] = g;
    const double t = HM ? mvh[i] * g : g;
    rv[i] = rv[i] + hk * t;
  }
  template <class F>
  __device__ __forceinline__ void grad(F&& f) {
    if (rows_done) __builtin_trap();
    rows_done = true;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (ok(u)) {
        const double gi = f(x[u], row(u));
        const double t = HM ? metric_of(u) * gi : gi;
        r[u] = r[u] + hk * t;
        if (DRIFT) x[u] = x[u] + hd * r[u];
        if (STORE && on && g_out)
          *reinterpret_cast<double*>(reinterpret_cast<char*>(g_out + (i64)G::off(u) * ld_out) + bo_out) = gi;
      }
      // bound the live temporaries (registers -> occupancy)
      if (DRIFT && LPC != 16 && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// What a proposal launch carries besides the trajectory (include/bkhip.h: bk_dr_proposal_funnel).
struct TrajArgs {
  const double* th_in; const double* rho_in; const double* g_in; i64 ld_in; const int32_t* idx;
  double* th_out; double* rho_out; double* g_out; double* logp_out; double* kin_out; i64 ld_out;
  const double* metric; double h; int steps; i64 n_host; i64 D; const uint32_t* n_dev; uint32_t* lanes_out;
  unsigned long long* lanes_total; double* H_out; double* hh_out; uint8_t* live_out; unsigned traj_blocks;
  const double* params;
  int hmc_first;  // the FIRST kick as bayes_kit/hmc.py:46,48 writes it: (rho + (-h/2) t) + h t, instead of rho + (h/2) t
  i64 auto_mid, auto_wide;  // geometry thresholds of a device-counted set (auto_geo())
};

// One whole delayed-rejection proposal (drghmc.py:319-346 -> :253-289) for n chains in ONE launch: gather chain
// idx[j] of the source point, first half-kick with the source's cached gradient + drift, (steps-1) x {gradient,
// kick, drift}, final gradient + log density, last half-kick, momentum flip, kinetic energy.  theta, rho never
// leave registers, and the sums over a chain's coordinates never leave the wavefront.
// LPC: lanes per chain; SL = slots per class = ceil((D-HEAD)/16) exactly; HM = a metric is given.  All
// compile-time: with generic sizes and a run-time metric flag the kernel needed 330 registers.
// A wavefront-step costs ~580 / ~700 / ~1100 cycles with 16 / 8 / 4 lanes per chain and serves 4 / 8 / 16
// chains; up to one wavefront per SIMD (1024 of them) the fewest cycles win, beyond that the fewest cycles per
// chain: 16 lanes per chain below AUTO_MID lanes, 8 below AUTO_WIDE, 4 from there on.
// (the thresholds travel with the launch -- TrajArgs -- so that tools/cfg4_geometry_scan.py can move them:
// BK_LANES_AUTO_MID / BK_LANES_AUTO_WIDE, read once per process)
constexpr i64 AUTO_MID = 4608, AUTO_WIDE = 12288;
struct AutoGeo { i64 mid, wide; };
static inline AutoGeo auto_geo() {
  static const AutoGeo g = []() {
    AutoGeo v = {AUTO_MID, AUTO_WIDE};
    if (const char* e = getenv("BK_LANES_AUTO_MID")) v.mid = atoll(e);
    if (const char* e = getenv("BK_LANES_AUTO_WIDE")) v.wide = atoll(e);
    if (v.mid < 1) v.mid = 1;
    if (v.wide < v.mid) v.wide = v.mid;
    return v;
  }();
  return g;
}

// Workgroups a launch needs for a set of at most n chains.  geo = 4 / 8 / 16: that many lanes per chain.  geo = 0 (the kernel
// picks the geometry from the count on the device): the most any admissible count needs -- fewer than AUTO_MID chains at 16 per
// workgroup, fewer than AUTO_WIDE at 32, n at 64 -- a quarter of sizing for 16 lanes per chain throughout (surplus workgroups
// exit after one scalar load and cost nothing measurable -- profiles/r5_cfg4_counted.md -- but fewer are never worse).
static inline unsigned blocks_for(i64 n, int geo, const AutoGeo& t) {
  if (geo != 0) return (unsigned)bk_cdiv(n, WAVES * (BK_WAVE / geo));
  const i64 a = bk_cdiv(n < t.mid ? n : t.mid - 1, WAVES * (BK_WAVE / 16));
  const i64 b = n < t.mid ? 0 : bk_cdiv(n < t.wide ? n : t.wide - 1, WAVES * (BK_WAVE / 8));
  const i64 c = n < t.wide ? 0 : bk_cdiv(n, WAVES * (BK_WAVE / 4));
  const i64 m = a > b ? (a > c ? a : c) : (b > c ? b : c);
  return (unsigned)m;
}

template <class DEN, int LPC, int SL, bool HM>
__device__ __forceinline__ void traj_body(const TrajArgs& a, i64 n, const bk_ghost_link& ghost, const bk_ghost0& g0,
                                          int lane, int wave) {
  constexpr int HEAD = DEN::HEAD;
  static_assert(HEAD >= 0 && HEAD <= MAX_HEAD, "0 <= HEAD <= 8");
  using G = Geo<LPC, SL>;
  constexpr int NU = G::NU, H1 = HEAD > 0 ? HEAD : 1;
  constexpr bool hm = HM;
  const i64 j0 = ((i64)blockIdx.x * WAVES + wave) * G::CHAINS;  // first chain of this wavefront
  // A whole WORKGROUP past the set leaves (uniform).  A wavefront past the set inside the set's last workgroup stays: its
  // lanes are off (they read chain 0 and store nothing), and it takes part in the workgroup's list appends (barriers).
  if ((i64)blockIdx.x * WAVES * G::CHAINS >= n) return;
  const int pos = lane & (LPC - 1);
  const i64 j = j0 + lane / LPC;
  const bool on = j < n;
  const bool writer = on && pos == 0;  // the lane that owns the head coordinates / the scalars
  const i64 src = on ? (a.idx ? (i64)a.idx[j] : j) : 0;
  const i64 ld_in = a.ld_in, ld_out = a.ld_out, D = a.D;
  const double h = a.h, half = 0.5 * a.h;
  const double* metric = a.metric;
  // this lane's rows: d(u) = HEAD + pos + off(u).  Byte offsets of its first row (host checked: < 2^32)
  const uint32_t bo_in = (uint32_t)(((i64)(HEAD + pos) * ld_in + src) * 8);
  const uint32_t bo_out = (uint32_t)(((i64)(HEAD + pos) * ld_out + (on ? j : 0)) * 8);
#define BKL_IN(p, u) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>((p) + (i64)G::off(u) * ld_in) + bo_in))
#define BKL_OUT(p, u) (*reinterpret_cast<double*>(reinterpret_cast<char*>((p) + (i64)G::off(u) * ld_out) + bo_out))
  bool tail_ok[G::KC];  // does the last slot of class k exist for this lane?
#pragma unroll
  for (int k = 0; k < G::KC; ++k) tail_ok[k] = HEAD + pos + G::off(k * SL + SL - 1) < D;
#define BKL_OK(u) (!G::last_slot(u) || tail_ok[(u) / SL])
  double x[NU], r[NU], mt[HM && LPC == 16 ? NU : 1];
  const bool regrad = a.g_in == nullptr;  // (uniform) the source's gradient is not cached: recomputed below
  // gather + first half-kick + drift (drghmc.py:276-278)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    // (lanes past the set read chain 0 -- src = 0 -- and compute on it; they store nothing.  No branch
    // around the loads of rows that exist for every lane.)
    const bool ok = BKL_OK(u);
    x[u] = ok ? BKL_IN(a.th_in, u) : 0.0;
    r[u] = ok ? BKL_IN(a.rho_in, u) : 0.0;
    const double mi = (hm && BKL_OK(u)) ? metric[HEAD + pos + G::off(u)] : 1.0;
    if (HM && LPC == 16) mt[u] = mi;
    if (!regrad) {
      double gin = ok ? BKL_IN(a.g_in, u) : 0.0;
      double t = hm ? mi * gin : gin;
      if (a.hmc_first) {
        r[u] = r[u] + (-half) * t;  // hmc.py:46
        r[u] = r[u] + h * t;        // hmc.py:48
      } else {
        r[u] = r[u] + half * t;
      }
      x[u] = x[u] + h * r[u];
    }
    // the gather is issued in batches of 8 rows: all 3*NU loads in flight at once would set the
    // kernel's register count (and so its occupancy for the whole trajectory)
    if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  double v[H1], rv[H1], mvh[H1], gv[H1];
#pragma unroll
  for (int i = 0; i < HEAD; ++i) {
    v[i] = a.th_in[(i64)i * ld_in + src];
    rv[i] = a.rho_in[(i64)i * ld_in + src];
    mvh[i] = hm ? metric[i] : 1.0;
    if (!regrad) {
      const double gin = a.g_in[(i64)i * ld_in + src];
      const double t = hm ? mvh[i] * gin : gin;
      if (a.hmc_first) {
        rv[i] = rv[i] + (-half) * t;
        rv[i] = rv[i] + h * t;
      } else {
        rv[i] = rv[i] + half * t;
      }
      v[i] = v[i] + h * rv[i];
    }
  }
  using StepCtx = TrajCtx<LPC, SL, HM, HEAD, false, true>;
  using KickCtx = TrajCtx<LPC, SL, HM, HEAD, false, false>;
  using EndCtx = TrajCtx<LPC, SL, HM, HEAD, true, false>;
  if (regrad) {
    // no cached gradient of the source point was handed over: evaluate it here and deliver it into the first half-kick +
    // drift (drghmc.py:276-278).  The same operations on the same values as the branch above -- the gradient is a function
    // of theta alone and its sums have one order -- for one evaluation more and one array less to read (and, at the other
    // end, to write and to scatter): the launches over all chains are memory launches.
    StepCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, half, h, pos, D, nullptr, 0, 0u, false, false};
    DEN::eval(c, a.params);
#pragma unroll
    for (int i = 0; i < HEAD; ++i) v[i] = v[i] + h * rv[i];
  }
  // (steps-1) x {gradient, kick, drift} (drghmc.py:280-283)
  for (int step = 0; step + 1 < a.steps; ++step) {
    StepCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, h, h, pos, D, nullptr, 0, 0u, false, false};
    DEN::eval(c, a.params);
#pragma unroll
    for (int i = 0; i < HEAD; ++i) v[i] = v[i] + h * rv[i];
  }
  // final gradient + log density (drghmc.py:285) and the last half-kick (:286)
  double logp_j;
  {
    EndCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, half, 0.0, pos, D, a.g_out, ld_out, bo_out, on, false};
    logp_j = DEN::eval(c, a.params);
    if (writer) {
      if (a.g_out) {
#pragma unroll
        for (int i = 0; i < HEAD; ++i) a.g_out[(i64)i * ld_out + j] = gv[i];
      }
      a.logp_out[j] = logp_j;
    }
  }
  // momentum flip (drghmc.py:345), kinetic energy (drghmc.py:250; same canonical order), outputs
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (BKL_OK(u)) {
      r[u] = -r[u];
      if (on) {
        BKL_OUT(a.rho_out, u) = r[u];
        BKL_OUT(a.th_out, u) = x[u];
      }
    }
  }
  double cs[G::KC];
  class_sums<G, SL>(cs, tail_ok, [&](int u) {
    return r[u] * (hm ? ((HM && LPC == 16) ? mt[(HM && LPC == 16) ? u : 0] : metric[HEAD + pos + G::off(u)]) * r[u] : r[u]);
  });
  const double ksum = reduce_lanes<LPC>(cs);
  double H_own = 0.0;
#pragma unroll
  for (int i = 0; i < HEAD; ++i) rv[i] = -rv[i];
  if (writer) {
    double hs = 0.0;
#pragma unroll
    for (int i = 0; i < HEAD; ++i) {
      const double mr = hm ? mvh[i] * rv[i] : rv[i];
      a.rho_out[(i64)i * ld_out + j] = rv[i];
      a.th_out[(i64)i * ld_out + j] = v[i];
      hs = i == 0 ? rv[i] * mr : hs + rv[i] * mr;
    }
    const double kin = 0.5 * (HEAD > 0 ? hs + ksum : ksum);
    a.kin_out[j] = kin;
    if (a.H_out) {
      // the level set-up of accept() (bk_dr_level_begin) for this lane, in the same launch:
      // H = -((-logp) + kin) (drghmc.py:421 -> :249-251); h and live follow below
      const double potential = -logp_j;
      const double Hj = -(potential + kin);
      H_own = Hj;
      a.H_out[j] = Hj;
    }
  }
  double h_own = 0.0;    // the produced level's h / live for this lane (h = 0, live = 1 unless its first ghost
  bool live_own = true;  // is run here as well)

  // ---- the proposal's FIRST GHOST, in the same wavefront (drghmc.py:424-436 with i = 0) ----------------------
  // Every lane of a level gets ghost 0, lane for lane, and a ghost of the first proposal kind has no ghosts of its
  // own: the wavefront that has just produced proposal P integrates P's ghost straight from its registers
  // (theta_P and the flipped momentum; the gradient at theta_P is evaluated again, the same values) --
  // no store + gather of the ghost's source, no ghost arrays at all (only its joint log density is ever used),
  // one launch instead of two.  Same operation sequence as a launch of its own.
  if (g0.steps > 0) {
    const double h2 = g0.h, half2 = 0.5 * g0.h;
    {  // first half-kick + drift from the proposal's end point (drghmc.py:276-278)
      StepCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, half2, h2, pos, D, nullptr, 0, 0u, false, false};
      DEN::eval(c, a.params);
#pragma unroll
      for (int i = 0; i < HEAD; ++i) v[i] = v[i] + h2 * rv[i];
    }
    for (int step = 0; step + 1 < g0.steps; ++step) {
      StepCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, h2, h2, pos, D, nullptr, 0, 0u, false, false};
      DEN::eval(c, a.params);
#pragma unroll
      for (int i = 0; i < HEAD; ++i) v[i] = v[i] + h2 * rv[i];
    }
    double logp_g;
    {
      KickCtx c{x, r, v, rv, gv, tail_ok, mt, mvh, metric, half2, 0.0, pos, D, nullptr, 0, 0u, false, false};
      logp_g = DEN::eval(c, a.params);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (BKL_OK(u)) r[u] = -r[u];
    class_sums<G, SL>(cs, tail_ok, [&](int u) {
      return r[u] * (hm ? ((HM && LPC == 16) ? mt[(HM && LPC == 16) ? u : 0] : metric[HEAD + pos + G::off(u)]) * r[u] : r[u]);
    });
    const double ksum_g = reduce_lanes<LPC>(cs);
    bool goes_on = false;
    if (writer) {
      double hs = 0.0;
#pragma unroll
      for (int i = 0; i < HEAD; ++i) {
        const double rr = -rv[i];
        const double mr = hm ? mvh[i] * rr : rr;
        hs = i == 0 ? rr * mr : hs + rr * mr;
      }
      const double kin_g = 0.5 * (HEAD > 0 ? hs + ksum_g : ksum_g);
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
    if (g0.next_index) bk_append_wg<WAVES>(goes_on, (int32_t)j, g0.next_index, g0.next_count);
  }
  if (writer && a.H_out) {
    a.hh_out[j] = h_own;
    a.live_out[j] = live_own ? 1 : 0;
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
    if (ghost.next_index) bk_append_wg<WAVES>(parent_goes_on, (int32_t)src, ghost.next_index, ghost.next_count);
  }
#undef BKL_IN
#undef BKL_OUT
#undef BKL_OK
}

// LPC_ARG = 4, 8 or 16: that geometry; 0: chosen at run time from the lane count, which only the device knows.
// The grid is sized by blocks_for(): what the largest admissible count of each geometry needs.
template <class DEN, int LPC_ARG, int SL, bool HM>
__global__ __launch_bounds__(BLOCK) void k_lane_traj(TrajArgs a, bk_scatter_job job, bk_ghost_link ghost, bk_ghost0 g0) {
  const int lane = threadIdx.x & (BK_WAVE - 1), wave = bk_wave_id();
  if (blockIdx.x >= a.traj_blocks) {
    // surplus workgroups: the previous stage's scatter (bk_scatter_job), one 64-lane unit per wavefront
    i64 jn = job.n;
    if (job.n_dev) {
      const i64 m = (i64)*job.n_dev;
      jn = m < jn ? m : jn;
    }
    const i64 ux_count = (job.n + 63) / 64;  // (units are laid out for the host-side bound)
    const i64 unit = ((i64)blockIdx.x - a.traj_blocks) * WAVES + wave;
    const i64 ux = unit % ux_count, uy = unit / ux_count;
    if (uy * BK_SCT_ROWS < job.D)
      bk_scatter_unit(ux, uy, lane, job.mask, job.index, jn, job.D, job.dst0, job.src0, job.dst1, job.src1, job.dst2,
                      job.src2, job.ld_dst, job.ld_src, job.sdst, job.ssrc);
    return;
  }
  // lanes actually in the set: read from device memory when the host only knows an upper bound
  i64 n = a.n_host;
  if (a.n_dev) {
    const i64 m = (i64)*a.n_dev;
    n = m < n ? m : n;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.lanes_out) *a.lanes_out = (uint32_t)n;
    if (a.lanes_total) *a.lanes_total += (unsigned long long)n;  // one writer per launch, launches are stream-ordered
    if (g0.steps > 0) {  // the fused first ghost runs over the same lanes
      if (g0.lanes_out) *g0.lanes_out = (uint32_t)n;
      if (g0.lanes_total) *reinterpret_cast<unsigned long long*>(g0.lanes_total) += (unsigned long long)n;
    }
  }
  if (LPC_ARG == 4 || (LPC_ARG == 0 && n >= a.auto_wide)) traj_body<DEN, 4, SL, HM>(a, n, ghost, g0, lane, wave);
  else if (LPC_ARG == 8 || (LPC_ARG == 0 && n >= a.auto_mid)) traj_body<DEN, 8, SL, HM>(a, n, ghost, g0, lane, wave);
  else traj_body<DEN, 16, SL, HM>(a, n, ghost, g0, lane, wave);
}

// Host side of bk_dr_proposal_funnel for any density (include/bkhip.h documents the arguments).
// SL_ONLY > 0: only that slot count is instantiated (a generated translation unit knows its D).
template <class DEN, int SL_ONLY = 0>
static int dr_proposal_launch(const double* theta_in, const double* rho_in, const double* grad_in, int64_t ld_in,
                              const int32_t* src_index, double* theta_out, double* rho_out, double* grad_out,
                              double* logp_out, double* kin_out, int64_t ld_out, const double* metric, double h,
                              int64_t steps, int64_t n, int64_t D, const uint32_t* n_dev, uint32_t* lanes_out,
                              uint64_t* lanes_total, double* H_out, double* h_out, uint8_t* live_out,
                              const bk_scatter_job* job_in, const bk_ghost_link* ghost_in, const bk_ghost0* g0_in,
                              const double* params, void* stream, bool hmc_first = false) {
  constexpr int HEAD = DEN::HEAD;
  // (grad_in NULL: the source's gradient is evaluated by the launch; grad_out NULL: the end point's is not stored)
  if (!theta_in || !rho_in || (!grad_in && hmc_first) || !theta_out || !rho_out || !logp_out || !kin_out ||
      steps < 1 || steps > 0x7fffffff || n < 0 || D < HEAD || D < 1)
    return BK_E_ARG;
  if (D - HEAD > MAX_ROWS) return BK_E_ARG;  // caller falls back to the step-by-step path
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
      job_blocks = (unsigned)bk_cdiv(bk_cdiv(job.n, 64) * bk_cdiv(job.D, BK_SCT_ROWS), WAVES);
  }
  // the kernel addresses a lane's rows with 32-bit byte offsets from wavefront-uniform row bases
  if ((ld_in > ld_out ? ld_in : ld_out) >= ((i64)1 << 32) / (8 * (CLASSES + HEAD))) return BK_E_ARG;
  hipStream_t s = bk_stream(stream);
  int need = (int)((D - HEAD + CLASSES - 1) / CLASSES);
  if (need < 1) need = 1;
  if (SL_ONLY > 0 && need != SL_ONLY) return BK_E_ARG;
  if (n == 0) {  // nothing to propose: a job still runs (every workgroup of the launch is a surplus one)
    if (g0.steps > 0 && g0.lanes_out) {
      int rc = (int)hipMemsetAsync(g0.lanes_out, 0, sizeof(uint32_t), s);
      if (rc != 0) return rc;
    }
    if (lanes_out) {
      int rc = (int)hipMemsetAsync(lanes_out, 0, sizeof(uint32_t), s);
      if (rc != 0) return rc;
    }
    if (!job_blocks) return BK_OK;
  }
  // Geometry.  A set known to be small -> 16 lanes per chain, 16 chains per workgroup; known to be large -> 4
  // lanes per chain, 64 chains per workgroup; a set whose size only the device knows (every set after the first
  // stage) -> decided by the kernel from *n_dev when the bound allows a large one.  Same values either way.
  // BK_FUNNEL_GEOMETRY=wide|mid|narrow overrides (wide = 4 lanes per chain).
  static const int forced = []() {
    const char* e = getenv("BK_FUNNEL_GEOMETRY");
    return !e ? 0 : (e[0] == 'n' ? 2 : e[0] == 'm' ? 3 : 1);  // wide | mid (8 lanes per chain) | narrow
  }();
  // 16: 16 lanes per chain; 4: 4 lanes per chain; 0: the kernel decides from *n_dev
  int geo;
  const AutoGeo thr = auto_geo();
  if (forced) geo = forced == 2 ? 16 : (forced == 3 ? 8 : 4);
  else if (n_dev) geo = n >= thr.mid ? 0 : 16;
  else geo = n >= thr.wide ? 4 : (n >= thr.mid ? 8 : 16);
  const unsigned traj_blocks = n == 0 ? 0u : blocks_for(n, geo, thr);
  const TrajArgs a = {theta_in, rho_in, grad_in, ld_in, src_index, theta_out, rho_out, grad_out, logp_out, kin_out,
                      ld_out, metric, h, (int)steps, n, D, n_dev, lanes_out,
                      reinterpret_cast<unsigned long long*>(lanes_total), H_out, h_out, live_out, traj_blocks, params,
                      hmc_first ? 1 : 0, thr.mid, thr.wide};
  dim3 grid(traj_blocks + job_blocks);
#define BKL_FT(LPC, R, M) k_lane_traj<DEN, LPC, R, M><<<grid, dim3(BLOCK), 0, s>>>(a, job, ghost, g0)
#define BKL_FT_ROWS(R)                 \
  do {                                 \
    if (geo == 16) {                   \
      if (metric) BKL_FT(16, R, true); \
      else BKL_FT(16, R, false);       \
    } else if (geo == 4) {             \
      if (metric) BKL_FT(4, R, true);  \
      else BKL_FT(4, R, false);        \
    } else if (geo == 8) {             \
      if (metric) BKL_FT(8, R, true);  \
      else BKL_FT(8, R, false);        \
    } else {                           \
      if (metric) BKL_FT(0, R, true);  \
      else BKL_FT(0, R, false);        \
    }                                  \
  } while (0)
  if constexpr (SL_ONLY > 0) {
    BKL_FT_ROWS(SL_ONLY);
  } else {
    switch (need) {  // slots per class, exactly: only a class's last slot can run past D
      case 1: BKL_FT_ROWS(1); break;
      case 2: BKL_FT_ROWS(2); break;
      case 3: BKL_FT_ROWS(3); break;
      case 4: BKL_FT_ROWS(4); break;
      case 5: BKL_FT_ROWS(5); break;
      case 6: BKL_FT_ROWS(6); break;
      case 7: BKL_FT_ROWS(7); break;
      default: BKL_FT_ROWS(8); break;
    }
  }
#undef BKL_FT_ROWS
#undef BKL_FT
  BK_RETURN_LAUNCH_STATUS();
}

// A whole HMC trajectory (bayes_kit/hmc.py:40-53) of every chain in ONE launch, for any lane-spread density: the same kernel
// with hmc.py's first kick.  In: theta, rho (overwritten with the end momentum, which HMC discards) and the cached gradient at
// theta; out: theta', the gradient and log density there, and the end point's kinetic energy 1/2 rho.(metric rho)
// (hmc.py:59 -> :37).  steps >= 1.
template <class DEN, int SL_ONLY = 0>
static int hmc_trajectory_launch(const double* theta_in, double* rho, const double* grad_in, int64_t ld_in, double* theta_out,
                                 double* grad_out, double* logp_out, double* kin_out, int64_t ld_out, const double* metric,
                                 double eps, int64_t steps, int64_t C, int64_t D, const double* params, void* stream) {
  return dr_proposal_launch<DEN, SL_ONLY>(theta_in, rho, grad_in, ld_in, nullptr, theta_out, rho, grad_out, logp_out, kin_out,
                                          ld_out, metric, eps, steps, C, D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, params, stream, true);
}

// ---- the same density as OPS on arrays in memory: gradient op and one-launch leapfrog step ---------------------------
// A per-step launch lives on its loads and stores, not on its arithmetic, so these kernels use the geometry that gives
// whole 512-byte row segments: LANE = CHAIN, a workgroup of 4 wavefronts serves 64 chains, wavefront w holds the class
// group w (classes w, w+4, w+8, w+12: the row set lane position w of the 4-lane trajectory geometry holds, so all the
// row bookkeeping is Geo<4, SL> with pos = w) and the four group sums meet in LDS:
//     q[w] = ((cs[w] + cs[w+4]) + cs[w+8]) + cs[w+12]     in wavefront w
//     s    = ((q[0] + q[1]) + q[2]) + q[3]                 in every wavefront, after one barrier
// -- the canonical order, hence the same bits as the trajectory kernel.  (Spreading a chain over 16 lanes, as the
// trajectory kernel does for sparse sets, makes every load and store a 32-byte piece: measured 7.6 us per gradient launch
// against 5.2 in config 4's counted draws.)  A workgroup whose chains lie past the count exits as a whole, before the
// barrier.  Successive sums alternate between two LDS buffers: one barrier per sum.
struct CoopRed {
  double (*part)[WAVES][BK_WAVE];  // [2][WAVES][64]
  int w, lane, k;
  __device__ __forceinline__ double operator()(const double* cs) {
    double(&buf)[WAVES][BK_WAVE] = part[k & 1];
    ++k;
    buf[w][lane] = ((cs[0] + cs[1]) + cs[2]) + cs[3];
    __syncthreads();
    return ((buf[0][lane] + buf[1][lane]) + buf[2][lane]) + buf[3][lane];
  }
};

// SL > 0: a wavefront's rows of its 64 chains sit in registers (one batch of loads, no second pass: on a small lane set a
// launch is a chain of memory round trips).  SL = 0: any D, the rows are walked in memory (sums, then gradient).
// STEP: the gradient is delivered INTO kick + drift, rho and theta rewritten in place (bk_leapfrog_step); else stored.
template <int SL, bool HM, int HEAD, bool STEP>
struct OpCtx {
  using G = Geo<4, (SL > 0 ? SL : 1)>;
  static constexpr int NU = SL > 0 ? G::NU : 1, KC = G::KC;
  double* x;             // [NU] rows (SL > 0)
  double* r;             // [NU] their momenta (STEP, SL > 0)
  const double* v;       // [HEAD]
  double* rv;            // [HEAD] (STEP)
  double* gv;            // [HEAD]
  const bool* tail_ok;   // [KC] (SL > 0)
  const double* mvh;     // [HEAD] head metric entries (STEP)
  const double* metric;  // device array or NULL (STEP)
  double* th;            // column of this lane's chain: th[d * ld]
  double* rho;           // the same of the momentum (STEP)
  double* g;             // the same of the gradient (NULL: none wanted; !STEP)
  i64 ld;
  double h;
  int pos;               // = wavefront index
  i64 D;
  bool on;
  bool want_lp;          // the launch was asked for the log density (!STEP)
  CoopRed red;
  bool rows_done;

  __device__ __forceinline__ i64 dims() const { return D; }
  __device__ __forceinline__ double head(int i) const { return v[i]; }
  __device__ __forceinline__ bool wants_logp() const { return !STEP && want_lp; }
  __device__ __forceinline__ i64 row(int u) const { return (i64)(HEAD + pos + G::off(u)); }
  __device__ __forceinline__ bool ok(int u) const { return !G::last_slot(u) || tail_ok[u / (SL > 0 ? SL : 1)]; }

  template <class F>
  __device__ __forceinline__ double sum(F&& f) {
    if (rows_done) __builtin_trap();  // (folds away: the flag is a compile-time constant after inlining)
    double cs[KC];
    if constexpr (SL > 0) {
      class_sums<G, SL>(cs, tail_ok, [&](int u) { return f(x[u], row(u)); });
    } else {
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        double acc = 0.0;
#pragma unroll 4
        for (i64 d = HEAD + pos + 4 * k; d < D; d += CLASSES) acc = acc + f(th[d * ld], d);
        cs[k] = acc;
      }
    }
    return red(cs);
  }
  __device__ __forceinline__ void grad_head(int i, double gval) {
    gv[i] = gval;
    if (STEP) {
      const double t = HM ? mvh[i] * gval : gval;
      rv[i] = rv[i] + h * t;
    }
  }
  template <class F>
  __device__ __forceinline__ void grad(F&& f) {
    if (rows_done) __builtin_trap();
    rows_done = true;
    if (!STEP && !g) return;
    if constexpr (SL > 0) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (ok(u)) {
          const double gi = f(x[u], row(u));
          if (STEP) {
            const double t = HM ? metric[row(u)] * gi : gi;
            r[u] = r[u] + h * t;
            x[u] = x[u] + h * r[u];
          } else if (on) {
            g[row(u) * ld] = gi;
          }
        }
    } else {
#pragma unroll 4
      for (i64 d = HEAD + pos; d < D; d += 4) {
        const double xd = th[d * ld];
        const double gi = f(xd, d);
        if (STEP) {
          const double t = HM ? metric[d] * gi : gi;
          const double rn = rho[d * ld] + h * t;
          if (on) {
            rho[d * ld] = rn;
            th[d * ld] = xd + h * rn;
          }
        } else if (on) {
          g[d * ld] = gi;
        }
      }
    }
  }
};

// gradient op (STEP = false: g and / or logp out) or one leapfrog step in place (STEP = true) for min(n, *n_dev) chains
template <class DEN, int SL, bool HM, bool STEP>
__global__ __launch_bounds__(BLOCK) void k_lane_op(double* th, double* rho, double* g, double* logp, i64 ld, const double* metric,
                                                   double h, const double* params, i64 n_host, i64 D, const uint32_t* n_dev) {
  constexpr int HEAD = DEN::HEAD;
  using C = OpCtx<SL, HM, HEAD, STEP>;
  using G = typename C::G;
  constexpr int H1 = HEAD > 0 ? HEAD : 1, SLq = SL > 0 ? SL : 1;
  __shared__ double part[2][WAVES][BK_WAVE];
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 n = bk_lanes(n_host, n_dev);
  if ((i64)blockIdx.x * BK_WAVE >= n) return;  // (whole workgroup past the set: uniform, before any barrier)
  const i64 j = (i64)blockIdx.x * BK_WAVE + lane;
  const bool on = j < n;
  const i64 col = on ? j : 0;  // (lanes past the set compute on chain 0 and store nothing: every lane meets the barrier)
  bool tail_ok[G::KC];
  double x[C::NU], r[C::NU];
  if constexpr (SL > 0) {
#pragma unroll
    for (int k = 0; k < G::KC; ++k) tail_ok[k] = HEAD + w + G::off(k * SLq + SLq - 1) < D;
#pragma unroll
    for (int u = 0; u < C::NU; ++u) {
      const bool ok = !G::last_slot(u) || tail_ok[u / SLq];
      const i64 o = (i64)(HEAD + w + G::off(u)) * ld + col;
      x[u] = ok ? th[o] : 0.0;
      if (STEP) r[u] = ok ? rho[o] : 0.0;
    }
  }
  double v[H1], rv[H1], mvh[H1], gv[H1];
#pragma unroll
  for (int i = 0; i < HEAD; ++i) {
    v[i] = th[(i64)i * ld + col];
    if (STEP) {
      rv[i] = rho[(i64)i * ld + col];
      mvh[i] = HM ? metric[i] : 1.0;
    }
  }
  C c{x, r, v, rv, gv, tail_ok, mvh, metric, th + col, STEP ? rho + col : nullptr, (!STEP && g) ? g + col : nullptr, ld, h, w, D, on,
      logp != nullptr, CoopRed{part, w, lane, 0}, false};
  const double lp = DEN::eval(c, params);
  // (a density without a sum() has no barrier of its own: every wavefront must have read the head coordinates before
  // wavefront 0 rewrites them)
  if (STEP && HEAD > 0) __syncthreads();
  if (!on) return;
  if (STEP) {
    if constexpr (SL > 0) {
#pragma unroll
      for (int u = 0; u < C::NU; ++u)
        if (!G::last_slot(u) || tail_ok[u / SLq]) {
          const i64 o = (i64)(HEAD + w + G::off(u)) * ld + j;
          rho[o] = r[u];
          th[o] = x[u];
        }
    }
    if (w == 0) {
#pragma unroll
      for (int i = 0; i < HEAD; ++i) {
        rho[(i64)i * ld + j] = rv[i];
        th[(i64)i * ld + j] = v[i] + h * rv[i];
      }
    }
  } else if (w == 0) {
    if (logp) logp[j] = lp;
    if (g) {
#pragma unroll
      for (int i = 0; i < HEAD; ++i) g[(i64)i * ld + j] = gv[i];
    }
  }
}

// slots per class for D, or 0 when the rows do not fit the registers
template <int HEAD>
static inline int op_slots(i64 D) {
  int need = (int)((D - HEAD + CLASSES - 1) / CLASSES);
  if (need < 1) need = 1;
  return need > MAX_SLOTS ? 0 : need;
}

#define BKL_OP_SWITCH(need, CALL) \
  switch (need) {                 \
    case 0: CALL(0); break;       \
    case 1: CALL(1); break;       \
    case 2: CALL(2); break;       \
    case 3: CALL(3); break;       \
    case 4: CALL(4); break;       \
    case 5: CALL(5); break;       \
    case 6: CALL(6); break;       \
    case 7: CALL(7); break;       \
    default: CALL(8); break;      \
  }

// Host side of the plugin ABI (bk_target_fn, and bk_target_fn_n with n_dev) for a lane-spread density.
// SL_ONLY >= 0: only that slot count is instantiated (a generated translation unit knows its D).
template <class DEN, int SL_ONLY = -1>
static int target_launch(const double* theta, double* grad, double* logp, int64_t ld, const double* params, int64_t C,
                         int64_t D, const uint32_t* n_dev, void* stream) {
  constexpr int HEAD = DEN::HEAD;
  if (!theta || (!grad && !logp) || C < 0 || D < HEAD || D < 1) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  const int need = op_slots<HEAD>(D);
  if (SL_ONLY >= 0 && need != SL_ONLY) return BK_E_ARG;
  const dim3 grid((unsigned)bk_cdiv(C, BK_WAVE)), block(BLOCK);
#define BKL_G(R)                                                                                                              \
  k_lane_op<DEN, R, false, false><<<grid, block, 0, s>>>(const_cast<double*>(theta), nullptr, grad, logp, ld, nullptr, 0.0, params, \
                                                         C, D, n_dev)
  if constexpr (SL_ONLY >= 0) {
    BKL_G(SL_ONLY);
  } else {
    BKL_OP_SWITCH(need, BKL_G)
  }
#undef BKL_G
  BK_RETURN_LAUNCH_STATUS();
}

// ---- one leapfrog step {gradient, kick, drift} (drghmc.py:280-283) as ONE launch, state in memory ---------------------
// For the step-by-step ("counted") path of a lane-spread density: theta and rho [D][ld] are advanced in place by one step of
// size h over min(n, *n_dev) chains -- half the launches of {gradient op, bk_leapfrog_kick_drift}, and the gradient never
// travels through memory.  Same arithmetic as those two launches (and as the trajectory kernel): bit-identical results.
// (SL = 0: every wavefront finishes its sums -- the barrier of the reduction -- before any row is rewritten, and a wavefront
// rewrites only the rows it alone reads.)
template <class DEN, int SL_ONLY = -1>
static int step_launch(double* theta, double* rho, int64_t ld, const double* metric, double h, const double* params, int64_t n,
                       int64_t D, const uint32_t* n_dev, void* stream) {
  constexpr int HEAD = DEN::HEAD;
  if (!theta || !rho || n < 0 || D < HEAD || D < 1) return BK_E_ARG;
  if (ld < n) return BK_E_ALIGN;
  if (n == 0) return BK_OK;
  hipStream_t s = bk_stream(stream);
  const int need = op_slots<HEAD>(D);
  if (SL_ONLY >= 0 && need != SL_ONLY) return BK_E_ARG;
  const dim3 grid((unsigned)bk_cdiv(n, BK_WAVE)), block(BLOCK);
#define BKL_S(R)                                                                                                               \
  do {                                                                                                                         \
    if (metric)                                                                                                                \
      k_lane_op<DEN, R, true, true><<<grid, block, 0, s>>>(theta, rho, nullptr, nullptr, ld, metric, h, params, n, D, n_dev);  \
    else                                                                                                                       \
      k_lane_op<DEN, R, false, true><<<grid, block, 0, s>>>(theta, rho, nullptr, nullptr, ld, metric, h, params, n, D, n_dev); \
  } while (0)
  if constexpr (SL_ONLY >= 0) {
    BKL_S(SL_ONLY);
  } else {
    BKL_OP_SWITCH(need, BKL_S)
  }
#undef BKL_S
  BK_RETURN_LAUNCH_STATUS();
}
#undef BKL_OP_SWITCH

}  // namespace bkl
