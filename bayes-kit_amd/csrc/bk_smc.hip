// Sequential Monte Carlo resampling (bayes_kit/smc.py:64-75): multinomial resampling of M particles from
// unnormalised weights with EXACTLY the arithmetic of the reference's
//     idxs = np.random.choice(M, size=M, replace=True, p=weights / weights.sum())          (smc.py:73)
// from the weights on, so that the ancestor indices are bit-identical given the same uniforms:
//   total = np.sum(weights)            numpy's pairwise summation, in pieces of 8,192 values added up in order
//   p_i   = weights_i / total
//   c_i   = c_{i-1} + p_i              np.cumsum: one sequential chain of rounded additions
//   idx_j = #{ i : c_i / c_{n-1} <= u_j }   (RandomState.choice: cdf /= cdf[-1]; searchsorted(u, side="right"))
// followed by the gather of the chosen particles' columns.  A chain of n dependent additions cannot be spread over
// lanes without changing its roundings, so the cdf is built by ONE wavefront: the 64 lanes stage tiles through LDS
// and sum the pairwise leaves, lane 0 runs the chains (n = 2,048: ~20 us; n = 65,536: ~0.5 ms).
#include "bk_common.hpp"

namespace {

constexpr int NP_SUM_PIECE = 8192;  // numpy's reduction buffer: np.sum adds the pairwise sums of such pieces in order
constexpr int NP_PW_BLOCK = 128;    // PW_BLOCKSIZE of numpy's pairwise sum
constexpr int MAX_LEAVES = 128;     // a piece splits into leaves of 65..128 values
constexpr int CDF_TILE = 4096;

// One piece (n <= 8192 values at a) summed as numpy's DOUBLE_pairwise_sum does; the whole wavefront calls it, the
// result is valid in lane 0.
__device__ double np_pairwise_piece(const double* a, int n, int* leaf_lo, int* leaf_n, double* leaf_sum, int* st_lo,
                                    int* st_n, int* st_flag, double* vals, int* n_leaves) {
  const int lane = threadIdx.x;
  if (n < 8) {  // (only a piece shorter than 8: numpy's plain loop, starting from 0.0)
    double res = 0.0;
    if (lane == 0)
      for (int i = 0; i < n; ++i) res = res + a[i];
    return res;
  }
  // 1) the leaves of the recursion (split at n/2 rounded down to a multiple of 8 while n > 128), left to right
  if (lane == 0) {
    int top = 0, cnt = 0;
    st_lo[0] = 0, st_n[0] = n, top = 1;
    while (top > 0) {
      --top;
      const int lo = st_lo[top], m = st_n[top];
      if (m <= NP_PW_BLOCK) {
        leaf_lo[cnt] = lo, leaf_n[cnt] = m, ++cnt;
      } else {
        int h = m / 2;
        h -= h % 8;
        st_lo[top] = lo + h, st_n[top] = m - h, ++top;  // right half: after the left one
        st_lo[top] = lo, st_n[top] = h, ++top;
      }
    }
    *n_leaves = cnt;
  }
  __syncthreads();
  const int L = *n_leaves;
  // 2) leaf sums, eight leaves at a time: lane j of a group of 8 owns numpy's running sum r[j] (a[j], a[8+j], ...),
  //    combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- additions commute, so the butterfly gives lane 0 of the
  //    group that very value -- then the up-to-7 left-over values in order
  for (int base = 0; base < L; base += 8) {
    const int leaf = base + (lane >> 3), j = lane & 7;
    const bool live = leaf < L;
    const int lo = live ? leaf_lo[leaf] : 0, m = live ? leaf_n[leaf] : 0;
    const int body = m - (m % 8);
    double r = 0.0;
    if (live) {
      r = a[lo + j];
      for (int i = 8; i < body; i += 8) r = r + a[lo + i + j];
    }
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    if (live && j == 0) {
      for (int i = body; i < m; ++i) r = r + a[lo + i];
      leaf_sum[leaf] = r;
    }
  }
  __syncthreads();
  // 3) the recursion's additions, left + right, in its own order (a post-order walk with a value stack)
  double res = 0.0;
  if (lane == 0) {
    int top = 0, vtop = 0, next = 0;
    st_n[0] = n, st_flag[0] = 0, top = 1;
    while (top > 0) {
      --top;
      const int m = st_n[top], flag = st_flag[top];
      if (m <= NP_PW_BLOCK) {
        vals[vtop++] = leaf_sum[next++];
      } else if (!flag) {
        int h = m / 2;
        h -= h % 8;
        st_n[top] = m, st_flag[top] = 1, ++top;
        st_n[top] = m - h, st_flag[top] = 0, ++top;
        st_n[top] = h, st_flag[top] = 0, ++top;
      } else {
        const double right = vals[--vtop], left = vals[--vtop];
        vals[vtop++] = left + right;
      }
    }
    res = vals[0];
  }
  __syncthreads();
  return res;
}

// cdf[i] = cumsum(w / np.sum(w))[i]  (NOT yet divided by its last entry: k_search divides on the fly)
__global__ __launch_bounds__(BK_WAVE) void k_choice_cdf(const double* w, double* cdf, i64 n) {
  __shared__ double tile[CDF_TILE];
  __shared__ double leaf_sum[MAX_LEAVES], vals[32];
  __shared__ int leaf_lo[MAX_LEAVES], leaf_n[MAX_LEAVES], st_lo[32], st_n[32], st_flag[32], n_leaves;
  __shared__ double total_s;
  const int lane = threadIdx.x;
  double total = 0.0;
  for (i64 lo = 0; lo < n; lo += NP_SUM_PIECE) {
    const int m = (int)(n - lo < NP_SUM_PIECE ? n - lo : NP_SUM_PIECE);
    const double piece = np_pairwise_piece(w + lo, m, leaf_lo, leaf_n, leaf_sum, st_lo, st_n, st_flag, vals, &n_leaves);
    total = lo == 0 ? piece : total + piece;
  }
  if (lane == 0) total_s = total;
  __syncthreads();
  total = total_s;
  double carry = 0.0;
  for (i64 lo = 0; lo < n; lo += CDF_TILE) {
    const int m = (int)(n - lo < CDF_TILE ? n - lo : CDF_TILE);
    for (int i = lane; i < m; i += BK_WAVE) tile[i] = w[lo + i] / total;
    __syncthreads();
    if (lane == 0) {
      double c = carry;
      int i = 0;
      for (; i + 8 <= m; i += 8) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = tile[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c = c + x[k];
          x[k] = c;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[i + k] = x[k];
      }
      for (; i < m; ++i) {
        c = c + tile[i];
        tile[i] = c;
      }
      carry = c;
    }
    __syncthreads();
    for (int i = lane; i < m; i += BK_WAVE) cdf[lo + i] = tile[i];
    __syncthreads();
  }
}

// idx[j] = number of entries of cdf / cdf[n-1] that are <= u[j]   (searchsorted side="right" on the normalised cdf)
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
  k_choice_cdf<<<dim3(1), dim3(BK_WAVE), 0, s>>>(weights, cdf_work, n);
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
