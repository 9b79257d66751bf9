// Global ranks of pooled draws (bayes_kit/rhat.py:27-59, `argsort().argsort() + 1`) without
// replicating them: the building blocks of a sample sort across ranks.
//   * bk_sort_by_key      stable radix sort of (double key, int64 payload) pairs -- rocPRIM's device
//                         radix sort compiled into this library (a vendor primitive, like a GEMM
//                         would be), everything around it is hand-written
//   * bk_count_below      bucket boundaries: for each splitter, the number of sorted keys below it
//   * bk_scatter_ranks    out[payload[j]] = base + j + 1
// The cross-rank choreography (samples -> splitters -> all_to_all of buckets -> ranks back) lives in
// bayes_kit_amd/diagnostics.py on torch.distributed (RCCL over xGMI; gloo in the CPU tests).
#include <cstring>

#include "bk_common.hpp"
#include <rocprim/device/device_radix_sort.hpp>

namespace {

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
  size_t bytes = 0;
  const double* k = nullptr;
  double* ko = nullptr;
  const long long* v = nullptr;
  long long* vo = nullptr;
  if (n <= 0) return 0;
  if (rocprim::radix_sort_pairs(nullptr, bytes, k, ko, v, vo, (size_t)n, 0, 64, (hipStream_t)0) != hipSuccess) return -1;
  return (int64_t)bytes;
}

int bk_sort_by_key(const double* keys_in, double* keys_out, const int64_t* vals_in, int64_t* vals_out, int64_t n,
                   void* work, int64_t work_bytes, void* stream) {
  if (n < 0 || (n > 0 && (!keys_in || !keys_out || !vals_in || !vals_out))) return BK_E_ARG;
  if (n == 0) return BK_OK;
  size_t bytes = (size_t)work_bytes;
  if (!work || work_bytes < bk_sort_by_key_work_bytes(n)) return BK_E_ARG;
  hipError_t e = rocprim::radix_sort_pairs(work, bytes, keys_in, keys_out, reinterpret_cast<const long long*>(vals_in),
                                           reinterpret_cast<long long*>(vals_out), (size_t)n, 0, 64,
                                           bk_stream(stream));
  return e == hipSuccess ? BK_OK : (int)e;
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
