// Shared helpers for libbkhip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkhip.h"

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
