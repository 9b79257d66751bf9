// RNG-consuming kernels: one chain per lane, each lane walks its own NumPy-compatible
// stream sequentially over d (the reference consumes `rng.normal(size=D)` in index order,
// bayes_kit/hmc.py:56, mala.py:44, drghmc.py:360-364).  Stores are coalesced across the
// 64 lanes of a wavefront because the state layout is chain-contiguous.
#include "bk_common.hpp"
#include "bk_rng.hpp"
#include "bk_welford.hpp"
#include "ziggurat_tables.inc"

namespace {

__device__ const uint64_t d_zig_ki[256] = BK_ZIG_KI_INIT;
__device__ const uint64_t d_zig_wi[256] = BK_ZIG_WI_BITS_INIT;
__device__ const uint64_t d_zig_fi[256] = BK_ZIG_FI_BITS_INIT;
const uint64_t h_zig_ki[256] = BK_ZIG_KI_INIT;
const uint64_t h_zig_wi[256] = BK_ZIG_WI_BITS_INIT;
const uint64_t h_zig_fi[256] = BK_ZIG_FI_BITS_INIT;

// 6 KiB of ziggurat tables staged in LDS: the table index is lane-divergent (8 random
// bits), so LDS (64 banks) serves it far better than the scalar/constant path.
struct ZigLds {
  uint64_t ki[256];
  double wi[256];
  double fi[256];
};

__device__ __forceinline__ void load_tables(ZigLds& t) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    t.ki[i] = d_zig_ki[i];
    t.wi[i] = bk::u64_as_double(d_zig_wi[i]);
    t.fi[i] = bk::u64_as_double(d_zig_fi[i]);
  }
  __syncthreads();
}

constexpr int RNG_BLOCK = 64;  // one wavefront per workgroup: C/64 workgroups spread over the CUs
// When C is too small to give every SIMD a wavefront (256 CUs x 4 SIMDs x 64 lanes = 65,536
// lanes), half-filled wavefronts (32 chains per workgroup) reach twice as many SIMDs: these
// kernels are latency bound, so that is faster (measured 103 -> 90 us at 32,768 chains).
static inline int rng_block(i64 C) { return C <= 32768 ? 32 : 64; }

__global__ __launch_bounds__(256) void k_init_philox(uint64_t* st, i64 ldr, uint64_t key0,
                                                     uint64_t chain0, i64 C) {
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  st[0 * ldr + c] = key0;
  st[1 * ldr + c] = chain0 + (uint64_t)c;
  for (int w = 2; w < 10; ++w) st[w * ldr + c] = 0;
  st[10 * ldr + c] = 4;
}

template <typename G>
__global__ __launch_bounds__(RNG_BLOCK) void k_refresh(uint64_t* st, i64 ldr, const double* loc_in,
                                                       double loc_mul, double scale, double* out,
                                                       i64 ld, const double* metric, double* kin_out,
                                                       const uint8_t* active, i64 C, i64 D) {
  __shared__ ZigLds tab;
  load_tables(tab);
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (active && !active[c]) return;
  G g;
  g.load(st, ldr, c);
  // kinetic energy summed in the same order as bk_leapfrog_finish (four contiguous quarters of
  // the dimensions, each sequential, combined ((p0+p1)+p2)+p3), so the same momentum has the
  // same energy whichever kernel computes it
  const i64 Dq = (D + 3) / 4;
  double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;
  auto emit = [&](i64 d, double z) {
    double loc = loc_in ? loc_in[d * ld + c] * loc_mul : 0.0;
    double v = loc + scale * z;  // numpy random_normal: loc + scale * z
    out[d * ld + c] = v;
    if (kin_out) {
      double mv = metric ? metric[d] * v : v;
      double t = v * mv;
      if (d < Dq) k0 = k0 + t;
      else if (d < 2 * Dq) k1 = k1 + t;
      else if (d < 3 * Dq) k2 = k2 + t;
      else k3 = k3 + t;
    }
  };
  i64 d = 0;
  for (; d + 1 < D; d += 2) {  // two normals per pass (same stream order, more ILP)
    bk::top_up(g);
    double z0, z1;
    bk::next_normal_pair(g, tab.ki, tab.wi, tab.fi, z0, z1);
    emit(d, z0);
    emit(d + 1, z1);
  }
  if (d < D) {
    bk::top_up(g);
    emit(d, bk::next_normal(g, tab.ki, tab.wi, tab.fi));
  }
  g.store(st, ldr, c);
  if (kin_out) kin_out[c] = 0.5 * (((k0 + k1) + k2) + k3);
}

template <typename G, bool LOG>
__global__ __launch_bounds__(RNG_BLOCK) void k_log_uniform(uint64_t* st, i64 ldr, double* out,
                                                           const uint8_t* active, i64 C) {
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (active && !active[c]) return;
  G g;
  g.load(st, ldr, c);
  double u = bk::next_double(g);
  out[c] = LOG ? log(u) : u;
  g.store(st, ldr, c);
}

template <typename G>
__global__ __launch_bounds__(RNG_BLOCK) void k_mala_propose(uint64_t* st, i64 ldr, const double* theta,
                                                            const double* grad, double* prop, i64 ld,
                                                            double eps, double s, i64 C, i64 D) {
  __shared__ ZigLds tab;
  load_tables(tab);
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  G g;
  g.load(st, ldr, c);
  i64 d = 0;
  for (; d + 1 < D; d += 2) {
    bk::top_up(g);
    double z0, z1;
    bk::next_normal_pair(g, tab.ki, tab.wi, tab.fi, z0, z1);
    i64 o = d * ld + c;
    prop[o] = (theta[o] + eps * grad[o]) + s * z0;  // mala.py:41-45, left to right
    prop[o + ld] = (theta[o + ld] + eps * grad[o + ld]) + s * z1;
  }
  if (d < D) {
    bk::top_up(g);
    i64 o = d * ld + c;
    prop[o] = (theta[o] + eps * grad[o]) + s * bk::next_normal(g, tab.ki, tab.wi, tab.fi);
  }
  g.store(st, ldr, c);
}

// ---- word-parallel ziggurat: LPC lanes of a wavefront per chain -------------------------------
// The one-lane-per-chain kernel above is latency bound: a chain's stream is consumed
// sequentially, ~2,200 cycles per normal with one wavefront per SIMD, and with few chains most
// SIMDs are idle (4096 chains = 64 wavefronts on 1024 SIMDs).  But Philox is counter based --
// word v of a stream is block(counter0 + v/4)[v%4] -- and 98.8 % of the normals use exactly
// one word.  So the LPC lanes of a segment evaluate 4 LPC consecutive words of ONE chain at
// once: every lane generates one block and runs the ziggurat fast test on its 4 words; the rare
// words that fail it get the complete slow path (wedge / tail) with positional access to the
// following words.  Which words actually START a normal is then resolved exactly in stream
// order: only the "exception" words can consume extra words, so a short scalar loop over the
// exceptions (a few per window, and only when they interact) tracks the covered range;
// everything else is ballot / scan arithmetic.  The result is bit for bit the sequential stream.
// Normals go to a chain-major scratch zt[c][d]; k_refresh_apply* turn them into the [D][C]
// layout with loc + scale*z.
//
// Round 6 (same stream, fewer instructions per pass; profiles/r6_generator.md):
//   * the slow path runs ONCE per pass for the lanes' first failing word (a second time in the 6 % of passes where
//     some lane has two) instead of once per word slot -- the wedge's exp() was executed 2.3 times per pass;
//   * the words of the window are staged in LDS (two 16-byte writes per lane): "the word after mine" is one LDS read
//     instead of eight ds_bpermute per pass and a select chain per access;
//   * ki / wi share one 16-byte table entry (one LDS read per word), (double)rabs is an exponent OR and a subtraction;
//   * a word's dimension comes from a DPP row scan of the lanes' emit counts with the running base folded into the
//     segment's first lane, instead of four ballots and sixteen masked popcounts; positions are window-relative ints;
//   * the end of the chain's last normal is broadcast once per launch, not once per pass.
struct SlowRes {
  double val;
  int len;   // words consumed by the attempt that starts at this word (>= 2)
  bool emit; // false: rejected wedge attempt (its words are consumed, no normal is produced)
};

struct __align__(16) ZigKW {
  uint64_t ki;
  double wi;
};
struct ZigLds2 {
  ZigKW kw[256];
  double fi[256];
};
constexpr int ZP_WAVES = 4;  // wavefronts per workgroup

__device__ __forceinline__ void load_tables2(ZigLds2& t) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    t.kw[i].ki = d_zig_ki[i];
    t.kw[i].wi = bk::u64_as_double(d_zig_wi[i]);
    t.fi[i] = bk::u64_as_double(d_zig_fi[i]);
  }
  __syncthreads();
}

// the ziggurat's word-local part (bk::zig_fast) with the table entry fetched in one read and the integer -> double
// conversion done by hand: rabs < 2^52, so 2^52 + rabs is the double with mantissa rabs and the subtraction is exact;
// the product is >= 0, so the sign (bit 8 of the word) is inserted, not xor-ed
__device__ __forceinline__ bool zig_fast2(uint64_t w, const ZigKW* kw, double& x) {
  const ZigKW e = kw[(uint32_t)w & 0xffu];
  const uint64_t rabs = (w >> 9) & 0x000fffffffffffffULL;
  const double xd = bk::u64_as_double(rabs | 0x4330000000000000ULL) - 4503599627370496.0;
  const uint64_t xm = bk::double_as_u64(xd * e.wi);
  const uint32_t hi = ((uint32_t)w << 23 & 0x80000000u) | (uint32_t)(xm >> 32);
  x = bk::u64_as_double(((uint64_t)hi << 32) | (uint32_t)xm);
  return rabs < e.ki;  // 99.2 %
}

// Positional access to the words that follow a lane's own: the segment's window is staged in LDS, `mine[j]` is word j
// counted from the lane's first.  NOTHING generates words beyond the window: an inlined "block after mine" in the wedge and
// tail branches is hoisted out of them by the compiler (common code of both) and every pass then pays for a second Philox
// block -- 160 v_mad_u64_u32 in the pass's main basic block instead of 80 (the round-5 kernel).  The window's last word and
// a tail that runs out of window are left to the next pass instead (zig_parallel_wave).
// The ziggurat's tail (idx == 0: one word in 4,000): pairs of the words that follow until the tail test passes, read
// from the segment's staged window (`mine` = the lane's first word there, `room` = words from it to the window's end).
// len = 0: the window ran out before the test passed -- the caller starts the next window at this attempt instead.
__device__ __forceinline__ SlowRes zig_tail_staged(const uint64_t* mine, int k, int room, uint64_t rabs) {
  const double zr = 3.6541528853610087963519472518, zinv = 0.27366123732975827203338247596;
  SlowRes r;
  r.val = 0.0, r.len = 0, r.emit = true;
  for (int j = k + 1; j + 1 < room; j += 2) {
    const double u1 = (double)(mine[j] >> 11) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(mine[j + 1] >> 11) * (1.0 / 9007199254740992.0);
    const double xx = -zinv * bk::bk_log1p(-u1);
    const double yy = -bk::bk_log1p(-u2);
    if (yy + yy > xx * xx) {
      r.val = ((rabs >> 8) & 1) ? -(zr + xx) : zr + xx;
      r.len = j + 2 - k;
      break;
    }
  }
  return r;
}

constexpr int DPP_ROW_SHR_ = 0x110, DPP_ROW_ROR_ = 0x120, DPP_WAVE_ROR1_ = 0x13C, DPP_ROW_BCAST15_ = 0x142,
              DPP_ROW_BCAST31_ = 0x143;

// inclusive scan of v over every segment of LPC lanes (rows of 16 lanes by DPP row shifts, then the row totals)
template <int LPC>
__device__ __forceinline__ int seg_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR_ + 1, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR_ + 2, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR_ + 4, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR_ + 8, 0xF, 0xF, false);
  if (LPC >= 32) v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15_, 0xA, 0xF, false);
  if (LPC >= 64) v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31_, 0xC, 0xF, false);
  return v;
}
// the value of the segment's LAST lane, delivered to its FIRST lane (other lanes: unspecified)
template <int LPC>
__device__ __forceinline__ int seg_last_to_first(int v) {
  if (LPC == 16) return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR_ + 1, 0xF, 0xF, false);
  if (LPC == 64) return __builtin_amdgcn_update_dpp(0, v, DPP_WAVE_ROR1_, 0xF, 0xF, false);
  return __shfl(v, ((int)(threadIdx.x & 63) | (LPC - 1)));
}

// LPC lanes per chain: a wavefront serves 64/LPC chains at once, each through a window of 4*LPC
// words per pass.  Every pass costs the same whatever part of it is used, so the window should divide
// the chain's ~1.008*D words with little left over: D = 1024 takes 5 passes of 256 words (the fifth
// for ~10 words) but 17 of 64 -- 4.25 wavefront passes per chain instead of 5 -- and D = 101 takes one
// 256-word pass per chain or two 64-word passes shared by four chains (half a wavefront pass per
// chain).  Fewer lanes per chain also mean fewer wavefronts, so small launches keep LPC = 64
// (zp_lanes_per_chain).  The stream is the same whatever LPC is: every quantity below that was
// wavefront-wide with LPC = 64 (ballots, ranks, the covered range) is per SEGMENT of LPC lanes.
template <int LPC>
__device__ __forceinline__ void zig_parallel_wave(uint64_t* st, i64 ldr, double* zt, i64 ldz, i64 C, i64 D,
                                                  uint64_t* snap, const ZigLds2& tab, uint64_t* win_lds, uint64_t* rk_lds,
                                                  i64 wave_index);

// The grid may be SMALLER than the number of chain groups (a "background" launch: bk_normals_chain_major with
// a bound on the workgroups): every workgroup then walks the groups blockIdx.x, blockIdx.x + gridDim.x, ...
// With one workgroup per CU the generator keeps one wavefront per SIMD and leaves the rest of every CU --
// registers, LDS, wavefront slots -- to a kernel that streams beside it.
template <int LPC>
__global__ __launch_bounds__(ZP_WAVES* BK_WAVE) void k_zig_parallel(uint64_t* st, i64 ldr, double* zt, i64 ldz,
                                                                    i64 C, i64 D, uint64_t* snap) {
  constexpr int G = BK_WAVE / LPC;  // chains per wavefront
  __shared__ ZigLds2 tab;
  __shared__ __align__(16) uint64_t win[ZP_WAVES][4 * BK_WAVE];
  __shared__ __align__(16) uint64_t rkeys[ZP_WAVES][G][20];
  load_tables2(tab);
  const i64 n_wg = (C + (i64)ZP_WAVES * G - 1) / ((i64)ZP_WAVES * G);
  for (i64 wg = blockIdx.x; wg < n_wg; wg += gridDim.x)
    zig_parallel_wave<LPC>(st, ldr, zt, ldz, C, D, snap, tab, win[bk_wave_id()], &rkeys[bk_wave_id()][0][0], wg * ZP_WAVES + bk_wave_id());
}

template <int LPC>
__device__ __forceinline__ void zig_parallel_wave(uint64_t* st, i64 ldr, double* zt, i64 ldz, i64 C, i64 D,
                                                  uint64_t* snap, const ZigLds2& tab, uint64_t* win_lds, uint64_t* rk_lds,
                                                  i64 wave_index) {
  constexpr int G = BK_WAVE / LPC;  // chains per wavefront
  constexpr int WW = 4 * LPC;       // words of a segment's window
  const int lane = threadIdx.x & (BK_WAVE - 1);
  const int seg = lane / LPC, l = lane % LPC;
  const i64 c = wave_index * G + seg;
  if ((c - seg) >= C) return;  // whole wavefront
  const bool live = c < C;     // (a wavefront's last segments may have no chain)
  const i64 cs = live ? c : C - 1;
  const unsigned long long segmask = (LPC == 64 ? ~0ULL : ((1ULL << (LPC & 63)) - 1)) << (seg * LPC);
  if (snap && live && l < BK_RNG_WORDS) snap[l * ldr + c] = st[l * ldr + c];  // the table before this call
  bk::Philox ph;
  ph.key0 = st[0 * ldr + cs];
  ph.key1 = st[1 * ldr + cs];
  const uint64_t k0 = st[2 * ldr + cs], k1 = st[3 * ldr + cs], k2 = st[4 * ldr + cs], k3 = st[5 * ldr + cs];
  const int v0 = (int)st[10 * ldr + cs];  // words of block `counter` already consumed (4 = all)
  const int Di = (int)D;
  // the chain's ten pairs of Philox round keys, in LDS for the whole launch (written by the segment's first ten lanes)
  uint64_t* const rk = rk_lds + 20 * seg;
  if (l < 10) {
    rk[2 * l] = ph.key0 + (uint64_t)l * 0x9E3779B97F4A7C15ULL;
    rk[2 * l + 1] = ph.key1 + (uint64_t)l * 0xBB67AE8584CAA73BULL;
  }
  __builtin_amdgcn_wave_barrier();
  // output addresses: a wave-uniform base (scalar registers) + a 32-bit offset per lane
  char* const zbase = reinterpret_cast<char*>(zt) + (size_t)((c - seg) * ldz) * sizeof(double);
  const uint32_t zoff = (uint32_t)((size_t)((cs - (c - seg)) * ldz) * sizeof(double));
  const uint64_t* const mine = win_lds + 4 * lane;  // the lane's four staged words, the segment's later ones behind them
  const int rel = 4 * l;           // window-relative position of the lane's first word
  int base = 0;                    // the segment's window starts at block `counter + base` ...
  int carry = v0;                  // ... whose first `carry` words an earlier attempt (or pass) has consumed
  int d_base = live ? 0 : Di;      // dimensions written so far (kept by the segment's first lane)
  int last_end = -1, last_base = 0;  // this lane holds the word that produced dimension D - 1: its attempt ends at the
                                     // window-relative position last_end of the pass that started at block last_base
  while (__any(l == 0 && d_base < Di)) {
    // 1. one Philox block per lane: block (counter + base + l), window-relative words 4 l + k
    uint64_t c0, c1, c2, c3, w0, w1, w2, w3;
    bk::Philox::ctr_add(k0, k1, k2, k3, (uint64_t)(base + l), c0, c1, c2, c3);
    bk::Philox::block_with_round_keys(rk, c0, c1, c2, c3, w0, w1, w2, w3);
    {
      ulonglong2* dst = reinterpret_cast<ulonglong2*>(win_lds + 4 * lane);
      dst[0] = make_ulonglong2(w0, w1);
      dst[1] = make_ulonglong2(w2, w3);
    }
    // 2. fast test on the four words
    double val[4];
    unsigned fail = 0;
    {
      const uint64_t wk[4] = {w0, w1, w2, w3};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (!zig_fast2(wk[k], tab.kw, val[k])) fail |= 1u << k;
    }
    // words consumed by an attempt of the previous window
    unsigned cov = 0;  // bit k: word k of this lane is consumed by an earlier attempt, or left to the next pass
    if (carry > rel) cov = carry - rel >= 4 ? 0xFu : (1u << (carry - rel)) - 1u;
    // The slow path reads "the word after mine" from the staged window, so the window's LAST word cannot start one:
    // when it fails the fast test (3 % of the passes) this pass leaves it alone and the next window starts at its
    // block -- one block of sixteen generated twice, no code that generates words on demand.
    int stop = WW;     // this pass handles the window-relative words below `stop` (the same in all lanes of a segment)
    const bool defer = l == LPC - 1 && (fail & 8u);
    if (defer) fail &= 7u, cov |= 8u;
    if (__ballot(defer) & segmask) stop = WW - 1;
    unsigned emitf = 0xFu;           // bit k: an attempt starting at word k produces a normal
    unsigned lenp = 0x01010101u;     // byte k: words that attempt consumes (1; a wedge 2; a tail 3, 5, ...)
    unsigned long long ex[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ex[k] = __ballot((fail >> k) & 1u);
    unsigned long long any = ex[0] | ex[1] | ex[2] | ex[3];
    int cover = carry;               // the words below this position are consumed
    if (any) {
      // 3. the full slow path, once for every lane's first failing word (again while some lane has another)
      __builtin_amdgcn_wave_barrier();
      bool tail_here = false;
      int short_at = WW;             // a tail attempt starting here ran out of window
      unsigned fm = fail;
      do {
        if (fm) {
          const int k = __ffs((int)fm) - 1;
          const uint64_t w = mine[k];
          const int idx = (int)((uint32_t)w & 0xffu);
          const uint64_t rabs = (w >> 9) & 0x000fffffffffffffULL;
          if (idx == 0) {  // the tail: one word in 4,000
            const SlowRes r = zig_tail_staged(mine, k, WW - rel, rabs);
            if (r.len == 0) {
              if (rel + k < short_at) short_at = rel + k;
            } else {
              if (r.len > 255) __builtin_trap();  // (cannot happen: WW <= 256; lenp holds bytes)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
                if (kk == k) val[kk] = r.val;
              lenp += (unsigned)(r.len - 1) << (8 * k);
            }
            tail_here = true;
          } else {         // the wedge: |x| again from the word (x * x does not see the sign), one more word
            const double xa = (bk::u64_as_double(rabs | 0x4330000000000000ULL) - 4503599627370496.0) * tab.kw[idx].wi;
            const double u = (double)(mine[k + 1] >> 11) * (1.0 / 9007199254740992.0);
            const bool emit = (tab.fi[idx - 1] - tab.fi[idx]) * u + tab.fi[idx] < exp(-0.5 * xa * xa);
            if (!emit) emitf &= ~(1u << k);
            lenp += 1u << (8 * k);
          }
          fm &= fm - 1;
        }
      } while (__any(fm != 0));
      // a tail that ran out of window (one pass in ~10^4): the pass stops in front of it, the next window starts at its
      // block.  A tail longer than a whole window (WW - 3 words: >= 30 rejected pairs in a row, p < 1e-40) cannot be
      // served this way and aborts loudly.
      const unsigned long long sm = __ballot(short_at < WW);
      if (sm) {
        if (sm & segmask) {
          const int q = __shfl(short_at, __ffsll((long long)(sm & segmask)) - 1);  // (positions ascend with the lanes)
          if (q < 4) __builtin_trap();
          if (q < stop) stop = q;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if (rel + kk >= q) fail &= ~(1u << kk), cov |= 1u << kk;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ex[k] = __ballot((fail >> k) & 1u);
        any = ex[0] | ex[1] | ex[2] | ex[3];
      }
      // 4. which words start an attempt.  The common window: every exception is a two-word attempt (wedge), none of
      //    them sits on a word that is already covered, no two are neighbours in the stream.  Then every exception
      //    starts an attempt, covers exactly the word behind it, and nothing has to be walked in order: a word is
      //    covered iff its predecessor in the segment is an exception (or the carry-in covers it).
      const bool odd = tail_here || (fail & cov) != 0;  // this lane holds an exception the shortcut cannot take
      // neighbours: words k, k+1 of one lane; word 3 of lane L and word 0 of lane L+1 of the same segment
      constexpr unsigned long long seg_last =  // lanes that are the last of their segment
          LPC == 64 ? 0x8000000000000000ULL : LPC == 32 ? 0x8000000080000000ULL : 0x8000800080008000ULL;
      const unsigned long long adj = (ex[0] & ex[1]) | (ex[1] & ex[2]) | (ex[2] & ex[3]) | (ex[3] & (ex[0] >> 1) & ~seg_last);
      const unsigned long long mine_mask = any & segmask;
      if (!__any(odd) && !adj) {
        if (mine_mask) {
          // covered: the word behind an exception
          cov |= (fail << 1) & 0xEu;
          if (l > 0 && ((ex[3] >> (lane - 1)) & 1ULL)) cov |= 1u;
          // the last exception of the segment ends the covered range
          const int Ll = 63 - __clzll((long long)mine_mask);  // last lane of the segment holding an exception
          int kl = 0;
#pragma unroll
          for (int k = 1; k < 4; ++k)
            if ((ex[k] >> Ll) & 1ULL) kl = k;
          cover = 4 * (Ll % LPC) + kl + 2;
        }
      } else {
        while (any) {
          const int L = __ffsll((long long)any) - 1;
          any &= any - 1;
          const bool mine_l = (L / LPC) == seg;
          const unsigned lens = (unsigned)__shfl((int)lenp, L);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if ((ex[k] >> L) & 1ULL) {
              const int pos = 4 * (L % LPC) + k;
              const int ln = (int)((lens >> (8 * k)) & 0xffu);
              if (mine_l && pos >= cover) {  // an actual attempt: it consumes words pos .. pos+ln-1
                cover = pos + ln;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                  if (rel + kk > pos && rel + kk < pos + ln) cov |= 1u << kk;
              }
            }
          }
        }
      }
    }
    // 5./6. emitting words, their dimension index, output
    const unsigned em = ~cov & emitf & 0xFu;
    const int cnt = __popc(em);
    const int incl = seg_inclusive_scan<LPC>(cnt + (l == 0 ? d_base : 0));
    int dim = incl - cnt;
    if (incl < Di) {  // every word of this lane comes before the chain's last dimension
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((em >> k) & 1u) *reinterpret_cast<double*>(zbase + (zoff + 8u * (uint32_t)(dim++))) = val[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((em >> k) & 1u) {
          if (dim < Di) *reinterpret_cast<double*>(zbase + (zoff + 8u * (uint32_t)dim)) = val[k];
          if (dim == Di - 1) last_end = rel + k + (int)((lenp >> (8 * k)) & 0xffu), last_base = base;
          ++dim;
        }
      }
    }
    d_base = seg_last_to_first<LPC>(incl);
    // the next window: behind this one, or at the block of the first word this pass left alone; an attempt that
    // started in this window may reach into the next one
    const int adv = stop >> 2;
    carry = (cover > stop ? cover : stop) - 4 * adv;
    base += adv;
  }
  // new stream position: p_end words into the block stream that starts at `counter`
  const unsigned long long lb = __ballot(last_end >= 0) & segmask;
  const i64 my_end = 4 * (i64)last_base + last_end;
  const i64 p_end = __shfl(my_end, lb ? __ffsll((long long)lb) - 1 : lane);
  if (l == 0 && live && lb && p_end > v0) {
    i64 bf = (p_end - 1) / 4;
    uint64_t a0, a1, a2, a3, b0, b1, b2, b3;
    bk::Philox::ctr_add(k0, k1, k2, k3, (uint64_t)bf, a0, a1, a2, a3);
    ph.block_at(a0, a1, a2, a3, b0, b1, b2, b3);
    st[2 * ldr + c] = a0; st[3 * ldr + c] = a1; st[4 * ldr + c] = a2; st[5 * ldr + c] = a3;
    st[6 * ldr + c] = b0; st[7 * ldr + c] = b1; st[8 * ldr + c] = b2; st[9 * ldr + c] = b3;
    st[10 * ldr + c] = (uint64_t)(p_end - 4 * bf);
  }
}

// Lanes per chain for a launch.  Measured (tools/zig_bench.py), a launch that fills the GPU costs
// passes x LPC per chain and nothing else -- 456 / 416 / 390 us at 65,536 x 1024 for LPC 64 / 32 / 16
// (5, 9, 17 passes), 72 / 44 / 38 us at 32,768 x 101 -- so among the choices that still give every SIMD
// several wavefronts the smallest product wins.  Small launches are latency bound: fewest passes
// first, then the fewest lanes (8.1 vs 9.6 us at 1000 x 70 with 32 instead of 64 lanes per chain).
static int zp_lanes_per_chain(i64 C, i64 D) {
  static const int forced = []() { const char* e = getenv("BK_ZP_LANES"); return e ? atoi(e) : 0; }();
  if (forced == 16 || forced == 32 || forced == 64) return forced;
  const double words = 1.0085 * (double)D + 6.0;  // a chain's D normals take ~0.85 % extra words
  const bool fills = C >= 8192;                    // LPC = 64 then gives >= 8 wavefronts per SIMD
  int best = 64;
  double best_cost = 0.0;
  for (int lpc = 64; lpc >= 16; lpc /= 2) {
    if (fills && C * lpc / 64 < 8192) break;
    const double passes = ceil(words / (4.0 * lpc));
    const double cost = fills ? passes * lpc : passes;
    if (lpc == 64 || cost <= best_cost) best = lpc, best_cost = cost;
  }
  return best;
}

// The generator with a SIDE JOB: of every `period` consecutive workgroups the last one runs a unit of a Welford update
// (bkw::welford_unit_v2: memory-bound) until its units are used up, the others generate (instruction-issue-bound): the two
// kinds are resident together and the launch takes about the longer of the two times (tools/zig_welford_overlap.py).
// A kernel of its own: k_zig_parallel -- the generator of every other caller -- is untouched.
struct DiagSide {
  double* mean; double* m2; const double* th; i64 ld, ld_th; const int64_t* n_dev; i64 n_off; i64 C2, D;
  // tracked series (k_record_series of bk_diag.hip): series[k][row][c] = theta[dims[k]][c], series[K][row][c] = logp[c]
  double* series; const int32_t* dims; int K; const double* logp; i64 cap, row_off, C;
  unsigned gx, w_units, rx, units, period;
};
template <int LPC>
__global__ __launch_bounds__(ZP_WAVES* BK_WAVE) void k_zig_parallel_side(uint64_t* st, i64 ldr, double* zt, i64 ldz,
                                                                         i64 C, i64 D, DiagSide job) {
  constexpr int G = BK_WAVE / LPC;  // chains per wavefront
  const unsigned b = blockIdx.x, k = b / job.period;
  if (b % job.period == job.period - 1 && k < job.units) {
    if (k < job.w_units) {
      const double n = (double)(*job.n_dev - job.n_off);
      bkw::welford_unit_v2<false>(k % job.gx, k / job.gx, (int)threadIdx.x, job.mean, job.m2, job.th, job.ld, job.ld_th, n,
                                  job.C2, job.D);
    } else {
      const unsigned u = k - job.w_units;
      const i64 c = (i64)(u % job.rx) * 256 + threadIdx.x;
      const int kk = (int)(u / job.rx);
      const i64 row = *job.n_dev - job.row_off;
      if (c < job.C && row >= 0 && row < job.cap)
        job.series[((i64)kk * job.cap + row) * job.C + c] = kk < job.K ? job.th[(i64)job.dims[kk] * job.ld_th + c] : job.logp[c];
    }
    return;
  }
  const unsigned before = (b + 1) / job.period;  // side workgroups among the blocks before this one
  const i64 wg = (i64)b - (before < job.units ? before : job.units);
  __shared__ ZigLds2 tab;
  __shared__ __align__(16) uint64_t win[ZP_WAVES][4 * BK_WAVE];
  __shared__ __align__(16) uint64_t rkeys[ZP_WAVES][G][20];
  load_tables2(tab);
  zig_parallel_wave<LPC>(st, ldr, zt, ldz, C, D, nullptr, tab, win[bk_wave_id()], &rkeys[bk_wave_id()][0][0], wg * ZP_WAVES + bk_wave_id());
}

// which parts of a side job the generator's launch can carry (the others become launches of their own)
static bool zig_side_welford(const bk_diag_job& j) {
  return j.mean && j.m2 && j.C > 0 && j.D > 0 && j.ld >= j.C && j.ld_theta >= j.C &&
         bkw::v2_applies(j.mean, j.m2, j.theta, j.ld, j.ld_theta, j.C) && !bk_streams_past_llc(3 * j.C * j.D);
}
static bool zig_side_record(const bk_diag_job& j) {
  return j.series && j.C > 0 && j.ld_theta >= j.C && j.K >= 0 && j.K + (j.logp ? 1 : 0) > 0 && (j.K == 0 || j.dims);
}

static void zig_parallel_side_launch(uint64_t* state, i64 ldr, double* zt, i64 ldz, i64 C, i64 D, const bk_diag_job& j,
                                     bool welford, bool record, hipStream_t s) {
  const int lpc = zp_lanes_per_chain(C, D);
  const i64 chains_per_wg = (i64)ZP_WAVES * (BK_WAVE / lpc);
  const i64 n_wg = bk_cdiv(C, chains_per_wg);
  DiagSide w = {};
  w.th = j.theta; w.ld_th = j.ld_theta; w.n_dev = j.n_dev;
  if (welford) {
    w.mean = j.mean; w.m2 = j.m2; w.ld = j.ld; w.n_off = j.n_offset; w.C2 = j.C / 2; w.D = j.D;
    w.gx = (unsigned)bk_cdiv(j.C / 2, 256);
    w.w_units = w.gx * (unsigned)bk_cdiv(j.D, bkw::EL_ROWS);
  }
  w.units = w.w_units;
  if (record) {
    w.series = j.series; w.dims = j.dims; w.K = (int)j.K; w.logp = j.logp; w.cap = j.capacity; w.row_off = j.row_offset; w.C = j.C;
    w.rx = (unsigned)bk_cdiv(j.C, 256);
    w.units += w.rx * (unsigned)(j.K + (j.logp ? 1 : 0));
  }
  const i64 total = n_wg + w.units;
  i64 period = total / w.units;  // (>= 1; the side units spread over the whole grid, or -- period 2 -- lead it)
  if (period < 2) period = 2;
  w.period = (unsigned)period;
  // (every side unit must have a slot: size the grid for the last one as well; generator blocks past the chains leave at once)
  i64 grid_n = total;
  const i64 last_side = (i64)(w.units - 1) * period + (period - 1);
  if (last_side >= grid_n) grid_n = last_side + 1;
  dim3 grid((unsigned)grid_n), block(ZP_WAVES * BK_WAVE);
  if (lpc == 16) k_zig_parallel_side<16><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, w);
  else if (lpc == 32) k_zig_parallel_side<32><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, w);
  else k_zig_parallel_side<64><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, w);
}

static void zig_parallel_launch(uint64_t* state, i64 ldr, double* zt, i64 ldz, i64 C, i64 D, uint64_t* snap,
                                hipStream_t s, i64 max_workgroups = 0) {
  const int lpc = zp_lanes_per_chain(C, D);
  const i64 chains_per_wg = (i64)ZP_WAVES * (BK_WAVE / lpc);
  i64 n_wg = bk_cdiv(C, chains_per_wg);
  if (max_workgroups > 0 && n_wg > max_workgroups) n_wg = max_workgroups;
  dim3 grid((unsigned)n_wg), block(ZP_WAVES * BK_WAVE);
  if (lpc == 16) k_zig_parallel<16><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, snap);
  else if (lpc == 32) k_zig_parallel<32><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, snap);
  else k_zig_parallel<64><<<grid, block, 0, s>>>(state, ldr, zt, ldz, C, D, snap);
}

// out[d][c] = loc + scale * zt[c][d]  (64 x 64 LDS tiles; zt is chain-major, out chain-contiguous)
__global__ __launch_bounds__(256) void k_refresh_apply(const double* zt, i64 ldz, const double* loc_in,
                                                       double loc_mul, double scale, double* out, i64 ld, i64 C,
                                                       i64 D) {
  __shared__ double tile[64][65];
  i64 c0 = (i64)blockIdx.x * 64, d0 = (i64)blockIdx.y * 64;
  int tx = threadIdx.x & 63, ty = bk_wave_id();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int cl = ty + 4 * i;
    i64 cc = c0 + cl, d = d0 + tx;
    tile[cl][tx] = (cc < C && d < D) ? zt[cc * ldz + d] : 0.0;
  }
  __syncthreads();
  i64 cc = c0 + tx;
  if (cc >= C) return;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int dl = ty + 4 * i;
    i64 d = d0 + dl;
    if (d < D) {
      double loc = loc_in ? loc_in[d * ld + cc] * loc_mul : 0.0;
      out[d * ld + cc] = loc + scale * tile[tx][dl];
    }
  }
}

// The same + the kinetic energy of the result (bk_leapfrog_finish with no kick: the four-quarter order of
// bk_integrator.hip) in ONE pass, optionally followed by the start of a DRGHMC draw for the chain
// (bk_dr_begin_retry): one launch and one sweep over rho instead of three launches and two sweeps.
// A workgroup serves 64 chains; wavefront w owns the quarter [w*Dq, (w+1)*Dq) of the dimensions, brings its
// [64 chains x 16 dims] pieces of the chain-major zt through a private LDS tile (lanes along d when reading zt,
// along the chains when writing rho) and sums v*(m*v) sequentially in d, as k_finish does.
struct DrBegin {
  const double* logp;
  double *H, *h, *rej;
  uint8_t* alive;
  double pr;
  uint32_t* counters;
  int n_counters;
  int64_t* draw_counter;
  uint64_t* state;
  i64 ldr;
};

constexpr int RK_CH = 16;  // dimensions per piece

template <bool BEGIN>
__global__ __launch_bounds__(256) void k_refresh_apply_kin(const double* zt, i64 ldz, const double* loc_in,
                                                           double loc_mul, double scale, double* out, i64 ld,
                                                           const double* metric, double* kin_out, i64 C, i64 D,
                                                           DrBegin b) {
  __shared__ double tile[4][64][RK_CH + 1];
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, w = bk_wave_id();
  const i64 c0 = (i64)blockIdx.x * 64, c = c0 + lane;
  const i64 Dq = (D + 3) / 4;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  if (BEGIN) {
    if (blockIdx.x == 0 && (int)threadIdx.x < b.n_counters) b.counters[threadIdx.x] = 0;
    if (b.draw_counter && blockIdx.x == 0 && threadIdx.x == 0) *b.draw_counter += 1;
  }
  const int sub = lane >> 4, dl = lane & 15;  // zt side: 4 chains x 16 dims per instruction
  double kin = 0.0;
  for (i64 p0 = 0; p0 < Dq; p0 += RK_CH) {  // (the same trip count in every wavefront: barriers inside)
    const i64 d0 = dlo + p0;
    double z[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      i64 cc = c0 + 4 * i + sub, d = d0 + dl;
      z[i] = (cc < C && d < dhi) ? zt[cc * ldz + d] : 0.0;
    }
    double loc[RK_CH];
#pragma unroll
    for (int u = 0; u < RK_CH; ++u)
      loc[u] = (loc_in && c < C && d0 + u < dhi) ? loc_in[(d0 + u) * ld + c] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[w][4 * i + sub][dl] = z[i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RK_CH; ++u) {
      if (c < C && d0 + u < dhi) {
        double lo = loc_in ? loc[u] * loc_mul : 0.0;
        double v = lo + scale * tile[w][lane][u];
        out[(d0 + u) * ld + c] = v;
        double mv = metric ? metric[d0 + u] * v : v;
        kin = kin + v * mv;
      }
    }
    __syncthreads();
  }
  part[w][lane] = kin;
  __syncthreads();
  if (w != 0 || c >= C) return;
  double s = part[0][lane];
#pragma unroll
  for (int k = 1; k < 4; ++k) s = s + part[k][lane];
  const double kc = 0.5 * s;
  kin_out[c] = kc;
  if (BEGIN) {
    // bk_dr_begin_retry for this chain (drghmc.py:365-371): the retry uniform comes after the chain's normals
    b.H[c] = dr_joint(b.logp[c], kc);
    b.h[c] = 0.0;
    const double r0 = 0.0;
    b.rej[c] = r0;
    bk::Philox g;
    g.load(b.state, b.ldr, c);
    double lu = log(bk::next_double(g));
    g.store(b.state, b.ldr, c);
    double retry = b.pr * r0;
    b.alive[c] = (lu < retry) ? 1 : 0;
  }
}

// k_refresh_apply_kin for SHORT rows (a chain's padded row of normals <= 120 doubles: config 4's 101).  The 64 chains of a
// workgroup are one contiguous piece of the chain-major zt: the wavefronts bring it in row by row (16-byte loads, a whole
// 832-byte row per instruction instead of four 128-byte pieces of four rows) into ONE LDS tile with an odd row pitch, and
// each wavefront then walks its quarter of the dimensions with lane = chain exactly as k_refresh_apply_kin does: the same
// operations in the same order on the same values.
template <bool BEGIN>
__global__ __launch_bounds__(256) void k_refresh_apply_kin_rows(const double* zt, i64 ldz, const double* loc_in,
                                                                double loc_mul, double scale, double* out, i64 ld,
                                                                const double* metric, double* kin_out, i64 C, i64 D,
                                                                DrBegin b) {
  extern __shared__ double rk_sm[];  // tile[64][ldz + 1], part[4][64]
  const int TS = (int)ldz + 1;
  double* part = rk_sm + 64 * TS;
  const int lane = threadIdx.x & 63, w = bk_wave_id();
  const i64 c0 = (i64)blockIdx.x * 64, c = c0 + lane;
  if (BEGIN) {
    if (blockIdx.x == 0 && (int)threadIdx.x < b.n_counters) b.counters[threadIdx.x] = 0;
    if (b.draw_counter && blockIdx.x == 0 && threadIdx.x == 0) *b.draw_counter += 1;
  }
  const int nch = (int)((C - c0 < 64) ? C - c0 : 64);
  if (2 * lane < ldz) {  // (ldz is a multiple of 8 and <= 128: a row is at most 64 double2)
#pragma unroll 4
    for (int cc = w; cc < nch; cc += 4) {
      const double2 v = *reinterpret_cast<const double2*>(zt + (c0 + cc) * ldz + 2 * lane);
      rk_sm[cc * TS + 2 * lane] = v.x;
      rk_sm[cc * TS + 2 * lane + 1] = v.y;
    }
  }
  __syncthreads();
  const i64 Dq = (D + 3) / 4;
  const i64 dlo = w * Dq, dhi = (dlo + Dq < D) ? dlo + Dq : D;
  double kin = 0.0;
  if (c < C) {
    for (i64 d0 = dlo; d0 < dhi; d0 += 8) {
      double loc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) loc[u] = (loc_in && d0 + u < dhi) ? loc_in[(d0 + u) * ld + c] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (d0 + u < dhi) {
          double lo = loc_in ? loc[u] * loc_mul : 0.0;
          double v = lo + scale * rk_sm[lane * TS + (int)(d0 + u)];
          out[(d0 + u) * ld + c] = v;
          double mv = metric ? metric[d0 + u] * v : v;
          kin = kin + v * mv;
        }
      }
    }
  }
  part[w * 64 + lane] = kin;
  __syncthreads();
  if (w != 0 || c >= C) return;
  double s = part[lane];
#pragma unroll
  for (int k = 1; k < 4; ++k) s = s + part[k * 64 + lane];
  const double kc = 0.5 * s;
  kin_out[c] = kc;
  if (BEGIN) {
    b.H[c] = dr_joint(b.logp[c], kc);
    b.h[c] = 0.0;
    const double r0 = 0.0;
    b.rej[c] = r0;
    bk::Philox g;
    g.load(b.state, b.ldr, c);
    double lu = log(bk::next_double(g));
    g.store(b.state, b.ldr, c);
    double retry = b.pr * r0;
    b.alive[c] = (lu < retry) ? 1 : 0;
  }
}

// launch of k_refresh_apply_kin / _rows
template <bool BEGIN>
static void refresh_apply_kin_launch(const double* work, i64 dp, const double* loc_in, double loc_mul, double scale,
                                     double* out, i64 ld, const double* metric, double* kin_out, i64 C, i64 D,
                                     const DrBegin& b, hipStream_t s) {
  const size_t lds = (size_t)(64 * (dp + 1) + 256) * sizeof(double);
  static const bool rows_off = getenv("BK_REFRESH_ROWS_OFF") != nullptr;  // (A/B with tools/cfg4_profile_run.py)
  if (dp <= 128 && lds <= 65536 && !rows_off)
    k_refresh_apply_kin_rows<BEGIN><<<dim3((unsigned)bk_cdiv(C, 64)), dim3(256), lds, s>>>(work, dp, loc_in, loc_mul, scale,
                                                                                          out, ld, metric, kin_out, C, D, b);
  else
    k_refresh_apply_kin<BEGIN><<<dim3((unsigned)bk_cdiv(C, 64)), dim3(256), 0, s>>>(work, dp, loc_in, loc_mul, scale, out,
                                                                                   ld, metric, kin_out, C, D, b);
}

// theta' = (theta + eps*grad) + s*z with z already drawn (mala.py:41-45), two rows per thread
__global__ __launch_bounds__(256) void k_mala_propose_z(const double* th, const double* g, const double* z,
                                                        double* prop, i64 ld, double eps, double s, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * 4;
  if (c >= C) return;
  double a[4], b[4], n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) {
      i64 o = (d0 + i) * ld + c;
      a[i] = th[o];
      b[i] = g[o];
      n[i] = z[o];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) prop[(d0 + i) * ld + c] = (a[i] + eps * b[i]) + s * n[i];
}

// the same with the normals chain-major (zt[c*ldz + d], as k_zig_parallel leaves them): 64x64
// tiles turned through LDS, so both sides stay coalesced
__global__ __launch_bounds__(256) void k_mala_propose_zt(const double* th, const double* g, const double* zt,
                                                         i64 ldz, double* prop, i64 ld, double eps, double s,
                                                         i64 C, i64 D) {
  __shared__ double tile[64][65];
  i64 c0 = (i64)blockIdx.x * 64, d0 = (i64)blockIdx.y * 64;
  int tx = threadIdx.x & 63, ty = bk_wave_id();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int cl = ty + 4 * i;
    i64 cc = c0 + cl, d = d0 + tx;
    tile[cl][tx] = (cc < C && d < D) ? zt[cc * ldz + d] : 0.0;
  }
  __syncthreads();
  i64 cc = c0 + tx;
  if (cc >= C) return;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int dl = ty + 4 * i;
    i64 d = d0 + dl;
    if (d < D) {
      i64 o = d * ld + cc;
      prop[o] = (th[o] + eps * g[o]) + s * tile[tx][dl];
    }
  }
}

// ---- ONE chain, ONE launch per MALA draw (the reference's own call shape: README.md:13-32) -----------------------
// A single chain driven by a host model cannot hide anything: its draw is a chain of small launches and copies
// (proposal, D2H, model call, H2D, proposal densities, accept test, select, D2H: 121 us).  Here the part of a draw
// that follows the model call -- proposal densities (mala.py:50-53, :68-79), accept test (metropolis.py:70-76), select
// (mala.py:62-64) -- and the NEXT draw's proposal (mala.py:41-45) are one launch of one lane; the model's outputs
// arrive in, and the draw and the next proposal leave through, host memory the device can address (pinned, mapped),
// and `seq` is written last so that the host can wait on it: 1 launch + 1 model call per draw.
// Stream order: this draw's uniform, then the next draw's D normals -- the reference's order; the table as it
// stands after the uniform (where the reference's generator stands between two sample() calls) goes to `snapshot`.
// Per-chain sums in the order of bk_mala_logq (four contiguous quarters, ((p0+p1)+p2)+p3).
template <typename G>
__global__ __launch_bounds__(64) void k_mala_single(uint64_t* st, i64 ldr, double* theta, double* grad, double* lp,
                                                    double* theta_prop, const double* host_in, double* host_out,
                                                    uint64_t* snapshot, i64 lds, uint8_t* mask, uint32_t* count,
                                                    double eps, double s2, i64 D, int have_prop, double seq) {
  __shared__ ZigLds tab;
  load_tables(tab);
  if (threadIdx.x != 0) return;
  G g;
  g.load(st, ldr, (i64)0);
  if (have_prop) {
    const double lp_p = host_in[0];
    const double* grad_p = host_in + 1;
    const i64 Dq = (D + 3) / 4;
    double pf[4] = {0.0, 0.0, 0.0, 0.0}, pr[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < 4; ++q) {
      const i64 dlo = q * Dq, dhi = dlo + Dq < D ? dlo + Dq : D;
      double sf = 0.0, sr = 0.0;
      for (i64 d = dlo; d < dhi; ++d) {
        const double a = theta[d], b = grad[d], p = theta_prop[d], qg = grad_p[d];
        const double xf = (p - a) - eps * b;  // mala.py:78
        const double xr = (a - p) - eps * qg;
        sf = sf + xf * xf;
        sr = sr + xr * xr;
      }
      pf[q] = sf;
      pr[q] = sr;
    }
    const double k = -0.25 / eps;  // mala.py:79
    const double fwd = k * (((pf[0] + pf[1]) + pf[2]) + pf[3]), rev = k * (((pr[0] + pr[1]) + pr[2]) + pr[3]);
    const double lu = log(bk::next_double(g));                     // metropolis.py:74
    const double l0 = lp[0];
    const bool acc = lu < (lp_p - l0) + (rev - fwd);               // metropolis.py:70-76 (strict <)
    if (acc) {
      for (i64 d = 0; d < D; ++d) {                                // mala.py:62-64
        theta[d] = theta_prop[d];
        grad[d] = grad_p[d];
      }
      lp[0] = lp_p;
      if (count) atomicAdd(count, 1u);
    }
    if (mask) mask[0] = acc ? 1 : 0;
    host_out[D] = acc ? lp_p : l0;                                 // mala.py:66
    host_out[D + 1] = acc ? 1.0 : 0.0;
    for (i64 d = 0; d < D; ++d) host_out[d] = theta[d];
  }
  if (snapshot) g.store(snapshot, lds, (i64)0);   // the logical stream position between two sample() calls
  for (i64 d = 0; d < D; ++d) {                                    // the next draw's proposal, mala.py:41-45
    const double z = bk::next_normal(g, tab.ki, tab.wi, tab.fi);
    const double p = (theta[d] + eps * grad[d]) + s2 * z;
    theta_prop[d] = p;
    host_out[D + 2 + d] = p;
  }
  g.store(st, ldr, (i64)0);
  __threadfence_system();
  *reinterpret_cast<volatile double*>(host_out + 2 * D + 2) = seq;  // last: the host waits for this
}

}  // namespace

extern "C" {

int bk_version(void) { return 100; }

int bk_mala_single_draw(int rng_kind, uint64_t* state, int64_t ldr, double* theta, double* grad, double* lp,
                        double* theta_prop, const double* host_in, double* host_out, uint64_t* snapshot,
                        int64_t lds, uint8_t* accept_mask, uint32_t* accept_count, double eps, double sqrt2eps,
                        int64_t D, int have_prop, double seq, void* stream) {
  if (!state || !theta || !grad || !lp || !theta_prop || !host_out || (have_prop && !host_in) || D < 0 || ldr < 1 ||
      (snapshot && lds < 1))
    return BK_E_ARG;
  hipStream_t s = bk_stream(stream);
  if (rng_kind == BK_RNG_PHILOX)
    k_mala_single<bk::Philox><<<dim3(1), dim3(64), 0, s>>>(state, ldr, theta, grad, lp, theta_prop, host_in, host_out,
                                                          snapshot, lds, accept_mask, accept_count, eps, sqrt2eps, D,
                                                          have_prop, seq);
  else if (rng_kind == BK_RNG_PCG64)
    k_mala_single<bk::Pcg64><<<dim3(1), dim3(64), 0, s>>>(state, ldr, theta, grad, lp, theta_prop, host_in, host_out,
                                                         snapshot, lds, accept_mask, accept_count, eps, sqrt2eps, D,
                                                         have_prop, seq);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_rng_init_philox(uint64_t* state, int64_t ldr, uint64_t key0, uint64_t chain_id0, int64_t C,
                       void* stream) {
  if (!state || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  k_init_philox<<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, bk_stream(stream)>>>(
      state, ldr, key0, chain_id0, C);
  BK_RETURN_LAUNCH_STATUS();
}

int64_t bk_refresh_work_elems(int64_t C, int64_t D) {
  int64_t dp = (D + 7) / 8 * 8;  // chain-major scratch zt[C][dp]
  return C * dp;
}

int bk_momentum_refresh(int rng_kind, uint64_t* state, int64_t ldr, const double* loc_in,
                        double loc_mul, double scale, double* out, int64_t ld, const double* metric,
                        double* kin_out, const uint8_t* active, int64_t C, int64_t D, double* work,
                        int64_t work_elems, void* stream) {
  if (!state || !out || C < 0 || D < 0 || ld < C || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  hipStream_t s = bk_stream(stream);
  if (rng_kind == BK_RNG_PHILOX) {
    // With scratch, enough dimensions to fill a wavefront's 256-word window, and few enough chains
    // that one-lane-per-chain would leave SIMDs idle or latency bound: one wavefront per chain.
    if (work && !active && D >= 32 && work_elems >= bk_refresh_work_elems(C, D)) {
      i64 dp = (D + 7) / 8 * 8;
      zig_parallel_launch(state, ldr, work, dp, C, D, nullptr, s);
      if (kin_out) {
        refresh_apply_kin_launch<false>(work, dp, loc_in, loc_mul, scale, out, ld, metric, kin_out, C, D, DrBegin{}, s);
      } else {
        dim3 g2((unsigned)bk_cdiv(C, 64), (unsigned)bk_cdiv(D, 64));
        k_refresh_apply<<<g2, dim3(256), 0, s>>>(work, dp, loc_in, loc_mul, scale, out, ld, C, D);
      }
      BK_RETURN_LAUNCH_STATUS();
    }
    k_refresh<bk::Philox><<<grid, block, 0, s>>>(state, ldr, loc_in, loc_mul, scale, out, ld, metric, kin_out, active,
                                                 C, D);
  } else if (rng_kind == BK_RNG_PCG64) {
    k_refresh<bk::Pcg64><<<grid, block, 0, s>>>(state, ldr, loc_in, loc_mul, scale, out, ld, metric, kin_out, active,
                                                C, D);
  } else {
    return BK_E_ARG;
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_refresh_begin(int rng_kind, uint64_t* state, int64_t ldr, const double* loc_in, double loc_mul, double scale,
                        double* out, int64_t ld, const double* metric, double* kin_out, int64_t C, int64_t D,
                        double* work, int64_t work_elems, const double* logp, double* cur_H, double* cur_h,
                        double* rej, uint8_t* alive, double prob_retry, uint32_t* counters, int64_t n_counters,
                        int64_t* draw_counter, const bk_diag_job* side, void* stream) {
  if (!state || !out || !kin_out || !logp || !cur_H || !cur_h || !rej || !alive || C < 0 || D < 0 || ld < C ||
      ldr < C || n_counters < 0 || n_counters > 64 || (n_counters > 0 && !counters))
    return BK_E_ARG;
  if (side && (!side->theta || !side->n_dev || side->C < 0 || side->D < 0 || (side->mean && !side->m2) ||
               (!side->mean && !side->series)))
    return BK_E_ARG;
  const bool with_generator = rng_kind == BK_RNG_PHILOX && work && D >= 32 && work_elems >= bk_refresh_work_elems(C, D);
  const bool ride_w = side && C > 0 && with_generator && zig_side_welford(*side);
  const bool ride_r = side && C > 0 && with_generator && zig_side_record(*side);
  // (what the generator's launch cannot carry: launches of their own, first -- they read the current point, the joint log
  // density and the draw count as this draw finds them)
  if (side && side->mean && !ride_w) {
    int rc = bk_welford_update_dev(side->mean, side->m2, side->ld, side->theta, side->ld_theta, side->n_dev, side->n_offset,
                                   side->C, side->D, stream);
    if (rc != BK_OK) return rc;
  }
  if (side && side->series && !ride_r) {
    int rc = bk_record_series_dev(side->theta, side->ld_theta, side->dims, side->K, side->logp, side->series, side->capacity,
                                  side->n_dev, side->row_offset, side->C, stream);
    if (rc != BK_OK) return rc;
  }
  if (C == 0) return BK_OK;
  if (with_generator) {
    hipStream_t s = bk_stream(stream);
    i64 dp = (D + 7) / 8 * 8;
    if (ride_w || ride_r) zig_parallel_side_launch(state, ldr, work, dp, C, D, *side, ride_w, ride_r, s);
    else zig_parallel_launch(state, ldr, work, dp, C, D, nullptr, s);
    DrBegin b = {logp, cur_H, cur_h, rej, alive, prob_retry, counters, (int)n_counters, draw_counter, state, ldr};
    refresh_apply_kin_launch<true>(work, dp, loc_in, loc_mul, scale, out, ld, metric, kin_out, C, D, b, s);
    BK_RETURN_LAUNCH_STATUS();
  }
  int rc = bk_momentum_refresh(rng_kind, state, ldr, loc_in, loc_mul, scale, out, ld, metric, kin_out, nullptr, C, D,
                               work, work_elems, stream);
  if (rc != BK_OK) return rc;
  return bk_dr_begin_retry(rng_kind, state, ldr, logp, kin_out, cur_H, cur_h, rej, alive, prob_retry, counters,
                           n_counters, draw_counter, C, stream);
}

int bk_log_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out, const uint8_t* active,
                   int64_t C, void* stream) {
  if (!state || !out || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_log_uniform<bk::Philox, true><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_log_uniform<bk::Pcg64, true><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out, const uint8_t* active, int64_t C,
               void* stream) {
  if (!state || !out || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_log_uniform<bk::Philox, false><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_log_uniform<bk::Pcg64, false><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_mala_propose(int rng_kind, uint64_t* state, int64_t ldr, const double* theta, const double* grad,
                    double* theta_prop, int64_t ld, double eps, double sqrt2eps, int64_t C, int64_t D,
                    void* stream) {
  if (!state || !theta || !grad || !theta_prop || C < 0 || D < 0 || ld < C || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_mala_propose<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, theta, grad, theta_prop,
                                                                     ld, eps, sqrt2eps, C, D);
  else if (rng_kind == BK_RNG_PCG64)
    k_mala_propose<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, theta, grad, theta_prop,
                                                                    ld, eps, sqrt2eps, C, D);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_mala_propose_from_normals(const double* theta, const double* grad, const double* z, int64_t z_stride_d,
                                 int64_t z_stride_c, double* theta_prop, int64_t ld, double eps,
                                 double sqrt2eps, int64_t C, int64_t D, void* stream) {
  if (!theta || !grad || !z || !theta_prop || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  if (z_stride_c == 1 && z_stride_d == ld) {
    dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, 4));
    k_mala_propose_z<<<grid, dim3(256), 0, bk_stream(stream)>>>(theta, grad, z, theta_prop, ld, eps, sqrt2eps, C, D);
  } else if (z_stride_d == 1 && z_stride_c >= D) {
    dim3 grid((unsigned)bk_cdiv(C, 64), (unsigned)bk_cdiv(D, 64));
    k_mala_propose_zt<<<grid, dim3(256), 0, bk_stream(stream)>>>(theta, grad, z, z_stride_c, theta_prop, ld, eps,
                                                                 sqrt2eps, C, D);
  } else {
    return BK_E_ALIGN;
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_normals_chain_major(int rng_kind, uint64_t* state, int64_t ldr, double* zt, int64_t ldz, int64_t C,
                              int64_t D, uint64_t* snapshot, int64_t max_workgroups, void* stream) {
  if (!state || !zt || C < 0 || D < 0 || ldr < C || ldz < D || max_workgroups < 0) return BK_E_ARG;
  if (rng_kind != BK_RNG_PHILOX) return BK_E_ARG;
  if (C == 0) return BK_OK;
  if (D == 0) {
    if (snapshot)
      return (int)hipMemcpy2DAsync(snapshot, ldr * sizeof(uint64_t), state, ldr * sizeof(uint64_t), C * sizeof(uint64_t),
                                   BK_RNG_WORDS, hipMemcpyDeviceToDevice, bk_stream(stream));
    return BK_OK;
  }
  zig_parallel_launch(state, ldr, zt, ldz, C, D, snapshot, bk_stream(stream), max_workgroups);
  BK_RETURN_LAUNCH_STATUS();
}

// ---- host self-test hooks: same source, host compilation ----------------------------
static void host_tables(const double** wi, const double** fi) {
  static double s_wi[256], s_fi[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; ++i) {
      s_wi[i] = bk::u64_as_double(h_zig_wi[i]);
      s_fi[i] = bk::u64_as_double(h_zig_fi[i]);
    }
    init = true;
  }
  *wi = s_wi;
  *fi = s_fi;
}

int bk_host_normals(int rng_kind, uint64_t* w, double* out, int64_t n) {
  if (!w || !out || n < 0) return BK_E_ARG;
  const double *wi, *fi;
  host_tables(&wi, &fi);
  if (rng_kind == BK_RNG_PHILOX) {
    bk::Philox g;
    g.load(w, (i64)1, (i64)0);
    i64 i = 0;
    for (; i + 1 < n; i += 2) bk::next_normal_pair(g, h_zig_ki, wi, fi, out[i], out[i + 1]);
    if (i < n) out[i] = bk::next_normal(g, h_zig_ki, wi, fi);
    g.store(w, (i64)1, (i64)0);
  } else if (rng_kind == BK_RNG_PCG64) {
    bk::Pcg64 g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_normal(g, h_zig_ki, wi, fi);
    g.store(w, (i64)1, (i64)0);
  } else {
    return BK_E_ARG;
  }
  return BK_OK;
}

int bk_host_uniforms(int rng_kind, uint64_t* w, double* out, int64_t n) {
  if (!w || !out || n < 0) return BK_E_ARG;
  if (rng_kind == BK_RNG_PHILOX) {
    bk::Philox g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_double(g);
    g.store(w, (i64)1, (i64)0);
  } else if (rng_kind == BK_RNG_PCG64) {
    bk::Pcg64 g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_double(g);
    g.store(w, (i64)1, (i64)0);
  } else {
    return BK_E_ARG;
  }
  return BK_OK;
}

double bk_host_log1p(double x) { return bk::bk_log1p(x); }
double bk_host_exp(double x) { return bk_exp(x); }

}  // extern "C"
