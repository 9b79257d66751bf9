// Experiment: bandwidth of tile-shaped access ([D][C] arrays, a workgroup owns W chains x R rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int64_t i64;
typedef double dvec2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: read NA arrays (sum), 1: write NA arrays, 2: read NA then write NA (phased), 3: interleaved copy
template <int PAIRS, int E, int NA, int MODE, bool NT, bool XR>
__global__ __launch_bounds__(512) void k_tile(double* a0, double* a1, double* a2, double* a3, double* o0, double* o1, double* o2,
                                              double* o3, i64 ld, i64 C, i64 D, double* sink) {
  constexpr int T = 512, ROWS = T / PAIRS, CH = 2 * PAIRS, SLAB = ROWS * E;
  double* in[4] = {a0, a1, a2, a3};
  double* out[4] = {o0, o1, o2, o3};
  const int t = threadIdx.x, j = t % PAIRS, r = t / PAIRS;
  i64 bid = blockIdx.x;
  const i64 nS = (D + SLAB - 1) / SLAB, nCB = gridDim.x / nS;
  i64 cbk, sl;
  if (XR && nS == 1) { const i64 per = nCB / 8; cbk = (bid % 8) * per + bid / 8; sl = 0; }
  else { cbk = bid / nS; sl = bid % nS; }
  const i64 c = cbk * CH + 2 * j;
  unsigned off[E];
#pragma unroll
  for (int e = 0; e < E; ++e) off[e] = (unsigned)(((sl * SLAB + r + ROWS * e) * ld + c) * 8);
  dvec2 v[NA][E];
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const dvec2* p = reinterpret_cast<const dvec2*>(reinterpret_cast<const char*>(in[k]) + off[e]);
        v[k][e] = NT ? __builtin_nontemporal_load(p) : *p;
      }
  } else {
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
      for (int e = 0; e < E; ++e) v[k][e] = dvec2{(double)t, (double)e};
  }
  if (MODE == 0) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
      for (int e = 0; e < E; ++e) s += v[k][e].x + v[k][e].y;
    if (s == 1234.5) sink[0] = s;
    return;
  }
  if (MODE == 2) __syncthreads();
#pragma unroll
  for (int k = 0; k < NA; ++k)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      dvec2* p = reinterpret_cast<dvec2*>(reinterpret_cast<char*>(out[k]) + off[e]);
      if (NT) __builtin_nontemporal_store(v[k][e], p); else *p = v[k][e];
    }
}

int main() {
  // PAD=<columns>: rows that many columns further apart than C (the row pitch off a power of two)
  const i64 C = 65536, D = 1024, LD = C + (getenv("PAD") ? atol(getenv("PAD")) : 0), n = LD * D;
  printf("row pitch %ld doubles\n", (long)LD);
  double* buf[8]; double* sink;
  for (auto& b : buf) { CK(hipMalloc(&b, n * 8)); CK(hipMemset(b, 0, n * 8)); }
  CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto&& launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    const int reps = 8;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-46s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); CK(hipGetLastError());
  };
  const double B = (double)n * 8;
#define RUN(PA, EE, NA, MO, NTT, XRR, bytes) timeit("P=" #PA " E=" #EE " NA=" #NA " MODE=" #MO " NT=" #NTT " XR=" #XRR, bytes, [&] { \
    constexpr int SLAB = (512 / PA) * EE; const i64 nS = (D + SLAB - 1) / SLAB, nCB = C / (2 * PA); \
    k_tile<PA, EE, NA, MO, NTT, XRR><<<dim3((unsigned)(nCB * nS)), dim3(512)>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], LD, C, D, sink); })
  // 16 chains x 1024 rows per workgroup (the k_mala_step shape), 2 arrays
  RUN(8, 16, 2, 0, true, false, 2 * B);
  RUN(8, 16, 2, 0, true, true, 2 * B);
  RUN(8, 16, 2, 1, true, false, 2 * B);
  RUN(8, 16, 2, 1, true, true, 2 * B);
  RUN(8, 16, 2, 1, false, true, 2 * B);
  RUN(8, 16, 2, 2, true, true, 4 * B);
  // 8 chains x 1024 rows per workgroup half the size (64-byte row pieces), two per CU
  RUN(4, 8, 2, 0, true, false, 2 * B);
  RUN(4, 8, 2, 1, true, false, 2 * B);
  RUN(4, 8, 2, 2, true, false, 4 * B);
  RUN(4, 8, 2, 2, true, true, 4 * B);
  RUN(4, 8, 4, 2, true, true, 8 * B);
  RUN(8, 16, 4, 2, true, true, 8 * B);
  // wider tiles with fewer rows (same 16K elements per array per workgroup)
  RUN(16, 16, 2, 1, true, false, 2 * B);
  RUN(32, 16, 2, 1, true, false, 2 * B);
  RUN(64, 16, 2, 1, true, false, 2 * B);
  RUN(256, 16, 2, 1, true, false, 2 * B);
  RUN(16, 16, 2, 0, true, false, 2 * B);
  RUN(32, 16, 2, 0, true, false, 2 * B);
  RUN(64, 16, 2, 0, true, false, 2 * B);
  RUN(256, 16, 2, 0, true, false, 2 * B);
  // fewer slots (smaller workgroup footprint, more workgroups per CU)
  RUN(8, 4, 2, 1, true, false, 2 * B);
  RUN(8, 4, 2, 0, true, false, 2 * B);
  RUN(8, 4, 2, 2, true, false, 4 * B);
  RUN(16, 4, 2, 2, true, false, 4 * B);
  RUN(64, 4, 2, 2, true, false, 4 * B);
  return 0;
}
