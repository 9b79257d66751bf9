// Sequential Monte Carlo resampling (bayes_kit/smc.py:64-75): multinomial resampling of M
// particles from unnormalised weights, as numpy's Generator.choice(p=...) does it -- inclusive
// cumulative sum, normalise by the total, one uniform per draw, index = searchsorted(cdf, u,
// side="right") -- followed by the gather of the chosen particles' columns.
#include "bk_common.hpp"

namespace {

constexpr int SCAN_BLOCK = 1024;

// inclusive scan of w[0..n) into cdf (one workgroup, chunked; wavefront shuffles + LDS)
__global__ __launch_bounds__(SCAN_BLOCK) void k_cumsum(const double* w, double* cdf, i64 n) {
  __shared__ double wave_tot[SCAN_BLOCK / BK_WAVE];
  __shared__ double carry;
  const int lane = threadIdx.x & (BK_WAVE - 1), wave = bk_wave_id();
  if (threadIdx.x == 0) carry = 0.0;
  __syncthreads();
  for (i64 start = 0; start < n; start += SCAN_BLOCK) {
    i64 i = start + threadIdx.x;
    double v = i < n ? w[i] : 0.0;
    // inclusive scan inside the wavefront
#pragma unroll
    for (int off = 1; off < BK_WAVE; off <<= 1) {
      double t = __shfl_up(v, off);
      if (lane >= off) v = v + t;
    }
    if (lane == BK_WAVE - 1) wave_tot[wave] = v;
    __syncthreads();
    double base = carry;
    for (int k = 0; k < wave; ++k) base = base + wave_tot[k];
    if (i < n) cdf[i] = base + v;
    __syncthreads();
    if (threadIdx.x == SCAN_BLOCK - 1) carry = base + v;
    __syncthreads();
  }
}

// idx[j] = number of cdf entries <= u[j] * total   (searchsorted side="right" on cdf/total)
__global__ __launch_bounds__(256) void k_search(const double* cdf, i64 n, const double* u, int32_t* idx, i64 m) {
  i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double total = cdf[n - 1];
  double x = u[j];
  i64 lo = 0, hi = n;
  while (lo < hi) {
    i64 mid = (lo + hi) >> 1;
    if (cdf[mid] / total <= x) lo = mid + 1;  // numpy normalises the cdf, then compares
    else hi = mid;
  }
  if (lo > n - 1) lo = n - 1;
  idx[j] = (int32_t)lo;
}

__global__ __launch_bounds__(256) void k_gather_cols(const int32_t* idx, const double* src, i64 lds_, double* dst,
                                                     i64 ldd, i64 m, i64 D) {
  i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * 4;
  if (j >= m) return;
  i64 s = idx[j];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) dst[(d0 + i) * ldd + j] = src[(d0 + i) * lds_ + s];
}

}  // namespace

extern "C" {

int bk_resample_indices(const double* weights, int64_t n, const double* u, int64_t m, double* cdf_work,
                        int32_t* idx_out, void* stream) {
  if (!weights || !u || !cdf_work || !idx_out || n < 1 || m < 0 || n > 0x7fffffff) return BK_E_ARG;
  hipStream_t s = bk_stream(stream);
  k_cumsum<<<dim3(1), dim3(SCAN_BLOCK), 0, s>>>(weights, cdf_work, n);
  if (m > 0) k_search<<<dim3((unsigned)bk_cdiv(m, 256)), dim3(256), 0, s>>>(cdf_work, n, u, idx_out, m);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_gather_columns(const int32_t* index, const double* src, int64_t ld_src, double* dst, int64_t ld_dst,
                      int64_t m, int64_t D, void* stream) {
  if (!index || !src || !dst || m < 0 || D < 0) return BK_E_ARG;
  if (ld_dst < m) return BK_E_ALIGN;
  if (m == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(m, 256), (unsigned)bk_cdiv(D, 4));
  k_gather_cols<<<grid, dim3(256), 0, bk_stream(stream)>>>(index, src, ld_src, dst, ld_dst, m, D);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
