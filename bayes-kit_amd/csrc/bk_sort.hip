// Global ranks of pooled draws (bayes_kit/rhat.py:27-59, `argsort().argsort() + 1`) without
// replicating them: the building blocks of a sample sort across ranks.
//   * bk_sort_by_key      stable radix sort of (double key, int64 payload) pairs, hand-written (below)
//   * bk_count_below      bucket boundaries: for each splitter, the number of sorted keys below it
//   * bk_scatter_ranks    out[payload[j]] = base + j + 1
// The cross-rank choreography (samples -> splitters -> all_to_all of buckets -> ranks back) lives in
// bayes_kit_amd/diagnostics.py on torch.distributed (RCCL over xGMI; gloo in the CPU tests).
//
// The sort: least-significant-digit radix sort, 8 passes of 8 bits over the order-preserving image of the keys
// (sign bit flipped for non-negative doubles, all bits for negative ones: -0.0 < +0.0, -inf first, +inf / nan
// last).  A pass is three launches:
//   k_sort_hist     one workgroup per tile of 4,096 keys: the tile's digit histogram (LDS atomics), stored
//                   digit-major [256][tiles]
//   k_sort_scan     one workgroup per digit: exclusive scan of its row of tile counts, digit totals; the last
//                   workgroup to finish turns the 256 totals into digit bases (and notes a pass in which every
//                   key has the same digit: the scatter then degenerates to a copy)
//   k_sort_scatter  one workgroup per tile: each wavefront walks its quarter of the tile 64 keys at a time and
//                   ranks them among the wavefront's earlier keys of the same digit with 8 ballots (the lanes
//                   that agree on all 8 bits are the peers; rank = popcount of the peers below, the lowest peer
//                   advances the wavefront's LDS counter for the digit) -- stable by construction; the tile is
//                   then laid out in digit order in LDS and written from there, so that consecutive lanes
//                   write consecutive addresses of a digit's run instead of 8-byte pieces all over the output.
// 40 bytes of traffic per key and pass (8 histogram + 16 in + 16 out).
#include <cstring>

#include "bk_common.hpp"

namespace {

typedef unsigned long long u64;

constexpr int SORT_THREADS = 256, SORT_ITEMS = 16, SORT_TILE = SORT_THREADS * SORT_ITEMS, SORT_BINS = 256;
constexpr int SORT_WAVES = SORT_THREADS / BK_WAVE, SORT_CHUNK = SORT_TILE / SORT_WAVES;

__device__ __forceinline__ u64 sort_image(double k) {
  const u64 b = (u64)__double_as_longlong(k);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double sort_preimage(u64 u) {
  const u64 b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)b);
}

template <bool FIRST>
__device__ __forceinline__ u64 sort_load(const void* keys, i64 i) {
  return FIRST ? sort_image(static_cast<const double*>(keys)[i]) : static_cast<const u64*>(keys)[i];
}

template <bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const void* keys, i64 n, int shift, uint32_t* tile_hist,
                                                            i64 n_tiles) {
  __shared__ uint32_t h[SORT_WAVES][SORT_BINS];
  const int t = threadIdx.x, w = bk_wave_id();
  for (int i = t; i < SORT_WAVES * SORT_BINS; i += SORT_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const i64 base = (i64)blockIdx.x * SORT_TILE;
  u64 k[SORT_ITEMS];  // (all of the thread's loads in flight before the first LDS atomic)
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const i64 idx = base + t + (i64)i * SORT_THREADS;
    k[i] = idx < n ? sort_load<FIRST>(keys, idx) : 0;
  }
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i)
    if (base + t + (i64)i * SORT_THREADS < n) atomicAdd(&h[w][(k[i] >> shift) & 255], 1u);
  __syncthreads();
  tile_hist[(i64)t * n_tiles + blockIdx.x] = ((h[0][t] + h[1][t]) + h[2][t]) + h[3][t];
}

// exclusive scan of one digit's row of tile counts (in place); totals[d]; the last workgroup: bases[d] and the
// "one digit holds every key" flag of the pass
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scan(uint32_t* tile_hist, i64 n_tiles, i64 n, uint32_t* totals,
                                                            uint32_t* bases, uint32_t* done, uint32_t* same) {
  __shared__ uint32_t wsum[SORT_WAVES];
  __shared__ uint32_t carry;
  __shared__ bool last;
  const int t = threadIdx.x, lane = t & 63, w = bk_wave_id();
  uint32_t* row = tile_hist + (i64)blockIdx.x * n_tiles;
  if (t == 0) carry = 0;
  __syncthreads();
  constexpr int PER = 8;  // consecutive entries per thread and round
  for (i64 c0 = 0; c0 < n_tiles; c0 += SORT_THREADS * PER) {
    const i64 i0 = c0 + (i64)t * PER;
    uint32_t v[PER], mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      v[k] = i0 + k < n_tiles ? row[i0 + k] : 0;
      mine += v[k];
    }
    uint32_t incl = mine;  // inclusive scan of the threads' sums inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (int k = 0; k < w; ++k) before += wsum[k];
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (i0 + k < n_tiles) row[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (t == SORT_THREADS - 1) carry = before + incl;
    __syncthreads();
  }
  if (t == 0) {
    totals[blockIdx.x] = carry;
    __threadfence();
    last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // 256 totals -> exclusive bases (one value per thread)
  const uint32_t v = __hip_atomic_load(&totals[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (int k = 0; k < w; ++k) before += wsum[k];
  bases[t] = before + incl - v;
  if (t == 0) {
    *same = 0;
    *done = 0;  // (ready for the next pass)
  }
  __syncthreads();
  if ((i64)v == n) *same = 1;
}

// Tile of a workgroup of the scatter: workgroups are handed to the 8 XCDs round-robin, so XCD x takes the x-th
// contiguous eighth of the tiles.  Consecutive tiles write consecutive runs of every digit; when they run on the
// same XCD at about the same time, the two partial cache lines where their runs meet merge in that XCD's L2
// instead of going out to memory as two masked writes.
__device__ __forceinline__ i64 sort_tile_of_block(i64 n_tiles) {
  const i64 per = (n_tiles + 7) / 8;
  return (i64)(blockIdx.x % 8) * per + (i64)(blockIdx.x / 8);
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const void* kin, const i64* vin, void* kout, i64* vout,
                                                               i64 n, int shift, const uint32_t* tile_off,
                                                               const uint32_t* bases, const uint32_t* same,
                                                               i64 n_tiles) {
  __shared__ u64 sk[SORT_TILE];
  __shared__ i64 sv[SORT_TILE];
  __shared__ uint32_t cnt[SORT_WAVES][SORT_BINS];
  __shared__ uint32_t dig_start[SORT_BINS];
  __shared__ i64 out_base[SORT_BINS];  // position in the output of the tile's first key of a digit, minus dig_start
  __shared__ uint32_t wtot[SORT_WAVES];
  const int t = threadIdx.x, lane = t & 63, w = bk_wave_id();
  const i64 tile = sort_tile_of_block(n_tiles);
  if (tile >= n_tiles) return;
  const i64 base = tile * SORT_TILE;
  const int m = (int)((n - base < SORT_TILE) ? n - base : SORT_TILE);  // keys of this tile
  if (*same) {
    // every key has the same digit in this pass: the pass is the identity
    for (int i = t; i < m; i += SORT_THREADS) {
      const u64 k = sort_load<FIRST>(kin, base + i);
      if (LAST) static_cast<double*>(kout)[base + i] = sort_preimage(k);
      else static_cast<u64*>(kout)[base + i] = k;
      vout[base + i] = vin[base + i];
    }
    return;
  }
  for (int i = t; i < SORT_WAVES * SORT_BINS; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  // wavefront w owns keys [w * SORT_CHUNK, (w + 1) * SORT_CHUNK) of the tile, 64 at a time in order
  u64 key[SORT_ITEMS];
  i64 val[SORT_ITEMS];
  uint32_t rank[SORT_ITEMS];
  const u64 below = (1ull << lane) - 1;
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int li = w * SORT_CHUNK + i * 64 + lane;
    const bool on = li < m;
    key[i] = on ? sort_load<FIRST>(kin, base + li) : ~0ull;
    val[i] = on ? vin[base + li] : 0;
  }
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int li = w * SORT_CHUNK + i * 64 + lane;
    const bool on = li < m;
    const uint32_t d = (uint32_t)(key[i] >> shift) & 255;
    u64 peers = __ballot(on);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const u64 mb = __ballot(bit);
      peers &= bit ? mb : ~mb;
    }
    uint32_t old = 0;
    const int leader = __ffsll((long long)peers) - 1;
    if (on && lane == leader) {
      old = cnt[w][d];
      cnt[w][d] = old + (uint32_t)__popcll(peers);
    }
    old = __shfl(old, leader < 0 ? 0 : leader);
    rank[i] = old + (uint32_t)__popcll(peers & below);
  }
  __syncthreads();
  // per digit (thread = digit): the wavefronts' counts become exclusive prefixes; digit starts inside the tile
  {
    uint32_t c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
    cnt[0][t] = 0;
    cnt[1][t] = c0;
    cnt[2][t] = c0 + c1;
    cnt[3][t] = (c0 + c1) + c2;
    const uint32_t v = ((c0 + c1) + c2) + c3;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < w; ++k) before += wtot[k];
    const uint32_t start = before + incl - v;
    dig_start[t] = start;
    out_base[t] = ((i64)bases[t] + (i64)tile_off[(i64)t * n_tiles + tile]) - (i64)start;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int li = w * SORT_CHUNK + i * 64 + lane;
    if (li < m) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & 255;
      const uint32_t pos = dig_start[d] + cnt[w][d] + rank[i];
      sk[pos] = key[i];
      sv[pos] = val[i];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int idx = t + i * SORT_THREADS;
    if (idx < m) {
      const u64 k = sk[idx];
      const uint32_t d = (uint32_t)(k >> shift) & 255;
      const i64 pos = out_base[d] + idx;
      if (LAST) static_cast<double*>(kout)[pos] = sort_preimage(k);
      else static_cast<u64*>(kout)[pos] = k;
      vout[pos] = sv[idx];
    }
  }
}

struct SortPlan {
  i64 n_tiles;
  size_t off_vals, off_hist, off_small, bytes;
};

SortPlan sort_plan(i64 n) {
  SortPlan p;
  p.n_tiles = bk_cdiv(n, SORT_TILE);
  const size_t arr = ((size_t)n * 8 + 255) / 256 * 256;
  p.off_vals = arr;                                        // [0, arr): keys in flight, [arr, 2 arr): payloads
  p.off_hist = 2 * arr;
  p.off_small = p.off_hist + ((size_t)p.n_tiles * SORT_BINS * 4 + 255) / 256 * 256;
  p.bytes = p.off_small + (2 * SORT_BINS + 2) * 4;         // totals, bases, done, same
  return p;
}

// number of keys strictly below q[i] (keys ascending): one thread per query, binary search
__global__ __launch_bounds__(64) void k_count_below(const double* keys, i64 n, const double* q, i64 m, i64* out) {
  i64 i = (i64)blockIdx.x * 64 + threadIdx.x;
  if (i >= m) return;
  const double v = q[i];
  i64 lo = 0, hi = n;
  while (lo < hi) {
    i64 mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  out[i] = lo;
}

__global__ __launch_bounds__(256) void k_scatter_ranks(const i64* payload, i64 n, double base, double* out) {
  i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  out[payload[j]] = base + (double)(j + 1);
}

}  // namespace

extern "C" {

int64_t bk_sort_by_key_work_bytes(int64_t n) {
  if (n <= 0) return 0;
  if (n > 0x7fffffff) return -1;
  return (int64_t)sort_plan(n).bytes;
}

int bk_sort_by_key(const double* keys_in, double* keys_out, const int64_t* vals_in, int64_t* vals_out, int64_t n,
                   void* work, int64_t work_bytes, void* stream) {
  if (n < 0 || n > 0x7fffffff || (n > 0 && (!keys_in || !keys_out || !vals_in || !vals_out))) return BK_E_ARG;
  if (n == 0) return BK_OK;
  const SortPlan p = sort_plan(n);
  if (!work || work_bytes < (int64_t)p.bytes || !bk_aligned16(work) || keys_in == keys_out || vals_in == vals_out)
    return BK_E_ARG;
  hipStream_t s = bk_stream(stream);
  char* base = static_cast<char*>(work);
  void* ktmp = base;
  i64* vtmp = reinterpret_cast<i64*>(base + p.off_vals);
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + p.off_hist);
  uint32_t* totals = reinterpret_cast<uint32_t*>(base + p.off_small);
  uint32_t *bases = totals + SORT_BINS, *done = bases + SORT_BINS, *same = done + 1;
  hipError_t e = hipMemsetAsync(done, 0, 2 * sizeof(uint32_t), s);
  if (e != hipSuccess) return (int)e;
  const dim3 tiles((unsigned)p.n_tiles), block(SORT_THREADS);
  const dim3 tiles8((unsigned)(8 * bk_cdiv(p.n_tiles, 8)));  // (k_sort_scatter: an eighth of the tiles per XCD)
  // in -> tmp -> out -> tmp -> ... : eight passes end in `out`
  const void* kin = keys_in;
  const i64* vin = vals_in;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 8 * pass;
    void* kout = (pass & 1) ? static_cast<void*>(keys_out) : ktmp;
    i64* vout = (pass & 1) ? vals_out : vtmp;
    if (pass == 0) k_sort_hist<true><<<tiles, block, 0, s>>>(kin, n, shift, hist, p.n_tiles);
    else k_sort_hist<false><<<tiles, block, 0, s>>>(kin, n, shift, hist, p.n_tiles);
    k_sort_scan<<<dim3(SORT_BINS), block, 0, s>>>(hist, p.n_tiles, n, totals, bases, done, same);
    if (pass == 0)
      k_sort_scatter<true, false><<<tiles8, block, 0, s>>>(kin, vin, kout, vout, n, shift, hist, bases, same, p.n_tiles);
    else if (pass == 7)
      k_sort_scatter<false, true><<<tiles8, block, 0, s>>>(kin, vin, kout, vout, n, shift, hist, bases, same, p.n_tiles);
    else
      k_sort_scatter<false, false><<<tiles8, block, 0, s>>>(kin, vin, kout, vout, n, shift, hist, bases, same, p.n_tiles);
    kin = kout;
    vin = vout;
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_count_below(const double* sorted_keys, int64_t n, const double* queries, int64_t m, int64_t* out, void* stream) {
  if (n < 0 || m < 0 || (m > 0 && (!queries || !out)) || (n > 0 && !sorted_keys)) return BK_E_ARG;
  if (m == 0) return BK_OK;
  k_count_below<<<dim3((unsigned)bk_cdiv(m, 64)), dim3(64), 0, bk_stream(stream)>>>(sorted_keys, n, queries, m, out);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_scatter_ranks(const int64_t* payload, int64_t n, double base, double* out, void* stream) {
  if (n < 0 || (n > 0 && (!payload || !out))) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_scatter_ranks<<<dim3((unsigned)bk_cdiv(n, 256)), dim3(256), 0, bk_stream(stream)>>>(payload, n, base, out);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
