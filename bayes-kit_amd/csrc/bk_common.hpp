// Shared helpers for libbkhip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkhip.h"
#include "../../include/bkhip_math.h"

typedef int64_t i64;

#define BK_WAVE 64  // CDNA wavefront width (hard-coded: gfx950 only)

#define BK_RETURN_LAUNCH_STATUS()              \
  do {                                         \
    hipError_t bk_e_ = hipGetLastError();      \
    return bk_e_ == hipSuccess ? BK_OK : (int)bk_e_; \
  } while (0)

static inline hipStream_t bk_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline i64 bk_cdiv(i64 a, i64 b) { return (a + b - 1) / b; }

static inline bool bk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Whether a kernel touching `elems` doubles streams well past the 256 MiB Infinity Cache
// (then non-temporal accesses pay) or can be served from it (then they hurt).
static inline bool bk_streams_past_llc(i64 elems) { return elems * 8 > ((i64)192 << 20); }

// Index of the calling wavefront inside its workgroup AS A SCALAR: threadIdx.x / 64 is the same for
// all 64 lanes, but the compiler only knows that if told -- otherwise every loop bound, row guard and
// address derived from it is per-lane vector work (selects, 64-bit VGPR address math).
__device__ __forceinline__ int bk_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)); }

// Lane count of a launch: `n` is what the host knows (an upper bound used to size the grid);
// with n_dev the number of lanes actually in the set is read from device memory -- written by an
// earlier kernel on the same stream (bk_compact_indices, the appending accept tests) -- so the host
// never has to read it back and a whole delayed-rejection draw can be captured as one hipGraph.
// Surplus threads exit.
__device__ __forceinline__ i64 bk_lanes(i64 n, const uint32_t* n_dev) {
  if (!n_dev) return n;
  const i64 m = (i64)*n_dev;
  return m < n ? m : n;
}

// ---- accepted columns -> the chains' current point (bk_scatter_columns) -----------------------------------
// One unit = 64 lanes (lane = chain of the compacted set) x BK_SCT_ROWS dimensions: the copy of an accepted
// column is spread over D / BK_SCT_ROWS units (one lane walking all D rows with dependent load -> store pairs
// took 22 us per launch at D = 101, whatever the number of accepted lanes).  Shared by k_scatter and by the
// trajectory kernel, whose surplus workgroups run the PREVIOUS stage's scatter beside the trajectories.
constexpr int BK_SCT_ROWS = 8;
__device__ __forceinline__ void bk_scatter_unit(i64 ux, i64 uy, int lane, const uint8_t* mask, const int32_t* idx, i64 n,
                                                i64 D, double* d0, const double* s0, double* d1, const double* s1,
                                                double* d2, const double* s2, i64 ldd, i64 lds, double* sd,
                                                const double* ss) {
  const i64 j = ux * 64 + lane;
  if (j >= n || !mask[j]) return;
  const i64 g = idx ? (i64)idx[j] : j;
  if (sd && uy == 0) sd[g] = ss[j];
  const i64 b = uy * BK_SCT_ROWS;
  double x0[BK_SCT_ROWS], x1[BK_SCT_ROWS], x2[BK_SCT_ROWS];
#pragma unroll
  for (int u = 0; u < BK_SCT_ROWS; ++u)
    if (b + u < D) {
      x0[u] = s0[(b + u) * lds + j];
      if (d1) x1[u] = s1[(b + u) * lds + j];
      if (d2) x2[u] = s2[(b + u) * lds + j];
    }
#pragma unroll
  for (int u = 0; u < BK_SCT_ROWS; ++u)
    if (b + u < D) {
      d0[(b + u) * ldd + g] = x0[u];
      if (d1) d1[(b + u) * ldd + g] = x1[u];
      if (d2) d2[(b + u) * ldd + g] = x2[u];
    }
}

// ---- delayed rejection: helpers shared by bk_dr.hip and the trajectory kernel -----------------------------
// Append `value` to list[0 .. *count) for the lanes with `take` set: one atomic per wavefront (ballot +
// popcount), positions inside the wavefront in lane order.  Every lane of the wavefront must call it.
// The ORDER of a list built this way depends on which wavefront's atomic lands first, so it differs from
// run to run; the lane sets, and every chain's values, do not (a chain's trajectory does not depend on the
// lane that integrates it, and the coordinate sums have a canonical order).
__device__ __forceinline__ void bk_append(bool take, int32_t value, int32_t* list, uint32_t* count) {
  const unsigned long long b = __ballot(take);
  if (b == 0) return;  // wavefront-uniform
  const int lane = threadIdx.x & (BK_WAVE - 1);
  const int leader = __ffsll((long long)b) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(b));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  if (take) list[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = value;
}

// The same for a whole WORKGROUP of NW wavefronts: ONE atomic per workgroup.  Atomics of different wavefronts on one
// counter are served one after the other, ~10 ns each on MI355X (device scope: eight L2s, so they resolve at the memory
// side): 2,048 wavefronts appending to one list cost 20 us, 512 workgroups 5 (profiles/r6_cfg4_exp.md).  EVERY thread of
// the workgroup must call it (two barriers); positions inside the workgroup in wavefront, then lane order.
template <int NW>
__device__ __forceinline__ void bk_append_wg(bool take, int32_t value, int32_t* list, uint32_t* count) {
  __shared__ uint32_t wave_count[NW];
  __shared__ uint32_t wg_base;
  const unsigned long long b = __ballot(take);
  const int lane = threadIdx.x & (BK_WAVE - 1), wave = (int)(threadIdx.x / BK_WAVE);
  if (lane == 0) wave_count[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += wave_count[w];
    wg_base = total ? atomicAdd(count, total) : 0u;
  }
  __syncthreads();
  if (take) {
    uint32_t base = wg_base;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wave) base += wave_count[w];
    list[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = value;
  }
}

// joint log density of (theta, rho) from log p(theta) and the kinetic energy (drghmc.py:249-251)
__device__ __forceinline__ double dr_joint(double logp, double kin) {
  double potential = -logp;
  return -(potential + kin);
}

// the acceptance log-probability of lane j against its current point p (drghmc.py:441-446)
__device__ __forceinline__ double dr_accept_logprob(double Hj, double cH, double ph, double ch, double pr) {
  const double frac = ((Hj - cH) + (ph - ch)) + (pr * ph - pr * ch);  // drghmc.py:441-445
  return frac < 0.0 ? frac : 0.0;                                      // min(0, frac), :446
}

