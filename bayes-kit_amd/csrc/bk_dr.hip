// Delayed-rejection (DRGHMC) stage helpers: per-chain scalar bookkeeping of the
// reference's recursive accept (bayes_kit/drghmc.py:348-446) as flat kernels over lane
// sets, plus the compaction / scatter that keep only the active chains in flight.
#include "bk_common.hpp"
#include "bk_rng.hpp"

namespace {

constexpr int SC_BLOCK = 256;

// ---- stable compaction: one workgroup, 16 wavefronts, 8 flags per thread per pass ---------
constexpr int CP_BLOCK = 1024;
constexpr int CP_ITEMS = 8;
__global__ __launch_bounds__(CP_BLOCK) void k_compact(const uint8_t* mask, i64 n_host, int32_t* idx,
                                                      uint32_t* count, const uint32_t* n_dev) {
  const i64 n = bk_lanes(n_host, n_dev);
  __shared__ uint32_t wave_cnt[CP_BLOCK / BK_WAVE];
  __shared__ uint32_t base;
  const int lane = threadIdx.x & (BK_WAVE - 1), wave = bk_wave_id();
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const bool aligned = (reinterpret_cast<uintptr_t>(mask) & 7u) == 0;
  for (i64 start = 0; start < n; start += (i64)CP_BLOCK * CP_ITEMS) {
    i64 i0 = start + (i64)threadIdx.x * CP_ITEMS;
    uint32_t bits = 0;  // bit k set <=> mask[i0 + k] != 0
    if (aligned && i0 + CP_ITEMS <= n) {
      uint64_t w = *reinterpret_cast<const uint64_t*>(mask + i0);
#pragma unroll
      for (int k = 0; k < CP_ITEMS; ++k) bits |= ((w >> (8 * k)) & 0xffu) ? (1u << k) : 0u;
    } else {
#pragma unroll
      for (int k = 0; k < CP_ITEMS; ++k)
        if (i0 + k < n && mask[i0 + k]) bits |= 1u << k;
    }
    uint32_t mine = (uint32_t)__popc(bits);
    // exclusive prefix of `mine` inside the wavefront
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < BK_WAVE; off <<= 1) {
      uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == BK_WAVE - 1) wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t off = base + (incl - mine);
    for (int w2 = 0; w2 < wave; ++w2) off += wave_cnt[w2];
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k)
      if (bits & (1u << k)) idx[off++] = (int32_t)(i0 + k);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w2 = 0; w2 < CP_BLOCK / BK_WAVE; ++w2) t += wave_cnt[w2];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base;
}

__device__ __forceinline__ double joint(double logp, double kin) { return dr_joint(logp, kin); }

__global__ __launch_bounds__(SC_BLOCK) void k_dr_begin(const double* logp, const double* kin, double* H,
                                                       double* h, double* rej, uint8_t* alive, i64 C,
                                                       const uint32_t* n_dev) {
  i64 c = (i64)blockIdx.x * SC_BLOCK + threadIdx.x;
  if (c >= bk_lanes(C, n_dev)) return;
  H[c] = joint(logp[c], kin[c]);
  h[c] = 0.0;
  if (rej) rej[c] = 0.0;
  alive[c] = 1;
}

template <typename G>
__global__ __launch_bounds__(64) void k_dr_retry(uint64_t* st, i64 ldr, const double* rej, double pr,
                                                 uint8_t* alive, i64 C) {
  i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C || !alive[c]) return;
  G g;
  g.load(st, ldr, c);
  double lu = log(bk::next_double(g));
  g.store(st, ldr, c);
  double retry = pr * rej[c];  // drghmc.py:317 (bool * float)
  if (!(lu < retry)) alive[c] = 0;  // drghmc.py:370-371
}

// Start of a draw + the first stage's retry test in one launch (k_dr_begin, then k_dr_retry for every
// chain: rej = 0, so the test always passes -- drghmc.py:366-371 -- but its uniform is drawn), and the
// draw's lane counters are zeroed for the appending kernels below.
template <typename G>
__global__ __launch_bounds__(64) void k_dr_begin_retry(uint64_t* st, i64 ldr, const double* logp, const double* kin,
                                                       double* H, double* h, double* rej, uint8_t* alive, double pr,
                                                       uint32_t* counters, int n_counters, int64_t* draw_counter,
                                                       i64 C) {
  if (blockIdx.x == 0 && (int)threadIdx.x < n_counters) counters[threadIdx.x] = 0;
  if (draw_counter && blockIdx.x == 0 && threadIdx.x == 0) *draw_counter += 1;  // (one writer; launches are stream-ordered)
  i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  H[c] = joint(logp[c], kin[c]);
  h[c] = 0.0;
  const double r0 = 0.0;
  rej[c] = r0;
  G g;
  g.load(st, ldr, c);
  double lu = log(bk::next_double(g));
  g.store(st, ldr, c);
  double retry = pr * r0;             // drghmc.py:317
  alive[c] = (lu < retry) ? 1 : 0;    // drghmc.py:370-371
}

__global__ __launch_bounds__(SC_BLOCK) void k_dr_ghost(const double* ga, const int32_t* sub, i64 m,
                                                       double* h, uint8_t* live, double* a,
                                                       const uint32_t* n_dev) {
  i64 j = (i64)blockIdx.x * SC_BLOCK + threadIdx.x;
  if (j >= bk_lanes(m, n_dev)) return;
  i64 p = sub ? (i64)sub[j] : j;
  double g = ga[j];
  if (g == 0.0) {  // drghmc.py:430-432
    a[p] = -INFINITY;
    live[p] = 0;
  } else {
    h[p] = h[p] + log1p(-exp(g));  // drghmc.py:434-435
  }
}

// accept probability of a GHOST level followed by the update of its parent level (k_dr_accept_prob +
// k_dr_ghost in one launch: ghost lane j has exactly one parent lane p, so nobody else touches p)
__global__ __launch_bounds__(SC_BLOCK) void k_dr_accept_prob_ghost(const double* H, const double* par_H,
                                                                   const double* h, double* par_h,
                                                                   const int32_t* sub, double pr,
                                                                   const uint8_t* live, double* a, i64 n,
                                                                   const uint32_t* n_dev, uint8_t* par_live,
                                                                   double* par_a, int32_t* next_idx,
                                                                   uint32_t* next_count) {
  i64 j = (i64)blockIdx.x * SC_BLOCK + threadIdx.x;
  const bool on = j < bk_lanes(n, n_dev);
  bool still_live = false;
  i64 p = 0;
  if (on) {
    p = sub ? (i64)sub[j] : j;
    double g;
    if (live[j]) {
      g = dr_accept_logprob(H[j], par_H[p], h[j], par_h[p], pr);
      a[j] = g;
    } else {
      g = a[j];  // (-inf: set when one of this lane's own ghosts was accepted with probability one)
    }
    if (g == 0.0) {  // drghmc.py:430-432
      par_a[p] = -INFINITY;
      par_live[p] = 0;
    } else {
      par_h[p] = par_h[p] + log1p(-exp(g));  // drghmc.py:434-435
      still_live = true;
    }
  }
  // the parent lanes that go on to their next ghost (drghmc.py:424): the lane set of that trajectory
  if (next_idx) bk_append_wg<SC_BLOCK / BK_WAVE>(still_live, (int32_t)p, next_idx, next_count);  // (next_idx: the same for all)
}

__global__ __launch_bounds__(SC_BLOCK) void k_dr_accept_prob(const double* H, const double* cur_H,
                                                             const double* h, const double* cur_h,
                                                             const int32_t* cidx, double pr,
                                                             const uint8_t* live, double* a, i64 n,
                                                             const uint32_t* n_dev) {
  i64 j = (i64)blockIdx.x * SC_BLOCK + threadIdx.x;
  if (j >= bk_lanes(n, n_dev) || !live[j]) return;
  i64 p = cidx ? (i64)cidx[j] : j;
  a[j] = dr_accept_logprob(H[j], cur_H[p], h[j], cur_h[p], pr);
}

// h / live given: the stage's accept probability (k_dr_accept_prob against the chain's current point) is
// evaluated here first, one launch less per stage; `a` is then an output as well
constexpr int AT_BLOCK = 256;  // (one list atomic per 256 lanes: bk_append_wg)
template <typename G>
__global__ __launch_bounds__(AT_BLOCK) void k_dr_accept_test(uint64_t* st, i64 ldr, const int32_t* cidx,
                                                       double* a, const double* H, i64 n,
                                                       double* cur_H, double* cur_h, double* rej,
                                                       uint8_t* alive, uint8_t* accepted, const uint32_t* n_dev,
                                                       const double* h, const uint8_t* live, double pr,
                                                       int32_t* next_idx, uint32_t* next_count) {
  i64 j = (i64)blockIdx.x * AT_BLOCK + threadIdx.x;
  const bool on = j < bk_lanes(n, n_dev);
  bool again = false;
  i64 c = 0;
  if (on) {
    c = cidx ? (i64)cidx[j] : j;
    G g;
    g.load(st, ldr, c);
    double lu = log(bk::next_double(g));
    if (h && live[j]) a[j] = dr_accept_logprob(H[j], cur_H[c], h[j], cur_h[c], pr);
    double aj = a[j];
    if (lu < aj) {  // drghmc.py:378-381
      accepted[j] = 1;
      cur_H[c] = H[j];
      alive[c] = 0;
    } else {  // drghmc.py:383-384
      accepted[j] = 0;
      double r = log1p(-exp(aj));
      rej[c] = r;
      cur_h[c] = cur_h[c] + r;
      if (next_idx) {
        // the NEXT stage's retry test for this chain (k_dr_retry; drghmc.py:369-371): its uniform is the
        // next value of the chain's stream, drawn here instead of by a launch of its own
        double lu2 = log(bk::next_double(g));
        double retry = pr * r;  // drghmc.py:317 (bool * float)
        if (lu2 < retry) again = true;
        else alive[c] = 0;
      }
    }
    g.store(st, ldr, c);
  }
  // the chains that propose again: the lane set of the next stage
  if (next_idx) bk_append_wg<AT_BLOCK / BK_WAVE>(again, (int32_t)c, next_idx, next_count);
}

// lane = chain of the compacted set, blockIdx.y = a block of BK_SCT_ROWS dimensions (bk_scatter_unit)
__global__ __launch_bounds__(64) void k_scatter(const uint8_t* mask, const int32_t* idx, i64 n, i64 D,
                                                double* d0, const double* s0, double* d1, const double* s1,
                                                double* d2, const double* s2, i64 ldd, i64 lds, double* sd,
                                                const double* ss, const uint32_t* n_dev) {
  bk_scatter_unit(blockIdx.x, blockIdx.y, threadIdx.x, mask, idx, bk_lanes(n, n_dev), D, d0, s0, d1, s1, d2, s2, ldd,
                  lds, sd, ss);
}

}  // namespace

extern "C" {

int bk_compact_indices(const uint8_t* mask, int64_t n, int32_t* idx_out, uint32_t* count_out,
                       const uint32_t* n_dev, void* stream) {
  if (!mask || !idx_out || !count_out || n < 0 || n > 0x7fffffff) return BK_E_ARG;
  k_compact<<<dim3(1), dim3(CP_BLOCK), 0, bk_stream(stream)>>>(mask, n, idx_out, count_out, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_begin(const double* logp, const double* kin, double* cur_H, double* cur_h, double* rej,
                uint8_t* alive, int64_t C, void* stream) {
  if (!logp || !kin || !cur_H || !cur_h || !rej || !alive || C < 0) return BK_E_ARG;
  if (C == 0) return BK_OK;
  k_dr_begin<<<dim3((unsigned)bk_cdiv(C, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(logp, kin, cur_H, cur_h,
                                                                                           rej, alive, C, nullptr);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_level_begin(const double* logp, const double* kin, double* H, double* h, uint8_t* live, int64_t n,
                      const uint32_t* n_dev, void* stream) {
  if (!logp || !kin || !H || !h || !live || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_dr_begin<<<dim3((unsigned)bk_cdiv(n, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(logp, kin, H, h, nullptr,
                                                                                           live, n, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_retry_test(int rng_kind, uint64_t* state, int64_t ldr, const double* rej, double prob_retry,
                     uint8_t* alive, int64_t C, void* stream) {
  if (!state || !rej || !alive || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(C, 64)), block(64);
  if (rng_kind == BK_RNG_PHILOX)
    k_dr_retry<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, rej, prob_retry, alive, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_dr_retry<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, rej, prob_retry, alive, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_ghost_update(const double* ga, const int32_t* sub_index, int64_t m, double* h, uint8_t* live,
                       double* a, const uint32_t* n_dev, void* stream) {
  if (!ga || !h || !live || !a || m < 0) return BK_E_ARG;
  if (m == 0) return BK_OK;
  k_dr_ghost<<<dim3((unsigned)bk_cdiv(m, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(ga, sub_index, m, h, live,
                                                                                           a, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_accept_prob(const double* H, const double* cur_H, const double* h, const double* cur_h,
                      const int32_t* cur_index, double prob_retry, const uint8_t* live, double* a, int64_t n,
                      const uint32_t* n_dev, void* stream) {
  if (!H || !cur_H || !h || !cur_h || !live || !a || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_dr_accept_prob<<<dim3((unsigned)bk_cdiv(n, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(
      H, cur_H, h, cur_h, cur_index, prob_retry, live, a, n, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

static int dr_accept_test(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index, double* a,
                          const double* H, int64_t n, double* cur_H, double* cur_h, double* rej, uint8_t* alive,
                          uint8_t* accepted, const uint32_t* n_dev, const double* h, const uint8_t* live, double pr,
                          int32_t* next_idx, uint32_t* next_count, void* stream) {
  if (!state || !a || !H || !cur_H || !cur_h || !rej || !alive || !accepted || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(n, AT_BLOCK)), block(AT_BLOCK);
  if (rng_kind == BK_RNG_PHILOX)
    k_dr_accept_test<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, chain_index, a, H, n, cur_H,
                                                                       cur_h, rej, alive, accepted, n_dev, h, live, pr,
                                                                       next_idx, next_count);
  else if (rng_kind == BK_RNG_PCG64)
    k_dr_accept_test<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, chain_index, a, H, n, cur_H,
                                                                      cur_h, rej, alive, accepted, n_dev, h, live, pr,
                                                                      next_idx, next_count);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_accept_test(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index, const double* a,
                      const double* H, int64_t n, double* cur_H, double* cur_h, double* rej, uint8_t* alive,
                      uint8_t* accepted, const uint32_t* n_dev, void* stream) {
  return dr_accept_test(rng_kind, state, ldr, chain_index, const_cast<double*>(a), H, n, cur_H, cur_h, rej, alive,
                        accepted, n_dev, nullptr, nullptr, 0.0, nullptr, nullptr, stream);
}

int bk_dr_accept_prob_test(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index, const double* H,
                           const double* h, const uint8_t* live, double* a, double prob_retry, int64_t n,
                           double* cur_H, double* cur_h, double* rej, uint8_t* alive, uint8_t* accepted,
                           const uint32_t* n_dev, void* stream) {
  if (!h || !live) return BK_E_ARG;
  return dr_accept_test(rng_kind, state, ldr, chain_index, a, H, n, cur_H, cur_h, rej, alive, accepted, n_dev, h,
                        live, prob_retry, nullptr, nullptr, stream);
}

int bk_dr_accept_prob_test_next(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index, const double* H,
                                const double* h, const uint8_t* live, double* a, double prob_retry, int64_t n,
                                double* cur_H, double* cur_h, double* rej, uint8_t* alive, uint8_t* accepted,
                                const uint32_t* n_dev, int32_t* next_index, uint32_t* next_count, void* stream) {
  if (!h || !live || !next_index || !next_count || next_index == chain_index) return BK_E_ARG;
  return dr_accept_test(rng_kind, state, ldr, chain_index, a, H, n, cur_H, cur_h, rej, alive, accepted, n_dev, h,
                        live, prob_retry, next_index, next_count, stream);
}

int bk_dr_begin_retry(int rng_kind, uint64_t* state, int64_t ldr, const double* logp, const double* kin,
                      double* cur_H, double* cur_h, double* rej, uint8_t* alive, double prob_retry,
                      uint32_t* counters, int64_t n_counters, int64_t* draw_counter, int64_t C, void* stream) {
  if (!state || !logp || !kin || !cur_H || !cur_h || !rej || !alive || C < 0 || ldr < C || n_counters < 0 ||
      n_counters > 64 || (n_counters > 0 && !counters))
    return BK_E_ARG;
  if (C == 0) return BK_OK;  // (no chain: no draw)
  dim3 grid((unsigned)bk_cdiv(C, 64)), block(64);
  if (rng_kind == BK_RNG_PHILOX)
    k_dr_begin_retry<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, logp, kin, cur_H, cur_h, rej, alive,
                                                                       prob_retry, counters, (int)n_counters, draw_counter, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_dr_begin_retry<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, logp, kin, cur_H, cur_h, rej, alive,
                                                                      prob_retry, counters, (int)n_counters, draw_counter, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_accept_prob_ghost(const double* H, const double* parent_H, const double* h, double* parent_h,
                            const int32_t* sub_index, double prob_retry, const uint8_t* live, double* a, int64_t n,
                            const uint32_t* n_dev, uint8_t* parent_live, double* parent_a, void* stream) {
  if (!H || !parent_H || !h || !parent_h || !live || !a || !parent_live || !parent_a || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_dr_accept_prob_ghost<<<dim3((unsigned)bk_cdiv(n, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(
      H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, n_dev, parent_live, parent_a, nullptr, nullptr);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_dr_accept_prob_ghost_next(const double* H, const double* parent_H, const double* h, double* parent_h,
                                 const int32_t* sub_index, double prob_retry, const uint8_t* live, double* a, int64_t n,
                                 const uint32_t* n_dev, uint8_t* parent_live, double* parent_a, int32_t* next_index,
                                 uint32_t* next_count, void* stream) {
  if (!H || !parent_H || !h || !parent_h || !live || !a || !parent_live || !parent_a || n < 0 || !next_index ||
      !next_count || next_index == sub_index)
    return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_dr_accept_prob_ghost<<<dim3((unsigned)bk_cdiv(n, SC_BLOCK)), dim3(SC_BLOCK), 0, bk_stream(stream)>>>(
      H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, n_dev, parent_live, parent_a, next_index, next_count);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_scatter_columns(const uint8_t* mask, const int32_t* index, int64_t n, int64_t D, double* dst0,
                       const double* src0, double* dst1, const double* src1, double* dst2, const double* src2,
                       int64_t ld_dst, int64_t ld_src, double* sdst, const double* ssrc, const uint32_t* n_dev,
                       void* stream) {
  if (!mask || !dst0 || !src0 || (dst1 && !src1) || (dst2 && !src2) || (sdst && !ssrc) || n < 0 || D < 0)
    return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_scatter<<<dim3((unsigned)bk_cdiv(n, 64), (unsigned)bk_cdiv(D > 0 ? D : 1, BK_SCT_ROWS)), dim3(64), 0, bk_stream(stream)>>>(
      mask, index, n, D, dst0, src0, dst1, src1, dst2, src2, ld_dst, ld_src, sdst, ssrc, n_dev);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
