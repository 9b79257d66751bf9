// What the fp64 matrix pipe sustains on gfx950, and what each ingredient of k_dense_apply's inner loop costs:
//   mode 0: 16 independent v_mfma_f64_16x16x4_f64 accumulators per wavefront, nothing else
//   mode 1: + the fragment reads from LDS (8 ds_read_b64 per k-slice, the GEMM's addresses)
//   mode 2: + a workgroup barrier every 4 k-slices (one "panel")
//   mode 3: + 12 LDS stores per panel (the next panel's stores; values that do not depend on the accumulators)
//   mode 4: + the panel's global loads (8 x 16 B per thread, issued at the top of the panel, consumed by the stores)
//   mode 5: + tile boundaries every 32 panels: 64 result stores per wavefront, then first panel load -> LDS -> barrier
// usage: mfma_rate_bench [workgroups_per_cu=2] [iters=2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int LDP = 144;

typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_rate(double* out, int iters, double seed, const double* src, double* dst,
                                                 size_t src_elems) {
  __shared__ double la[2][16][LDP], lb[2][16][LDP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = (w >> 1) * 64, wc = (w & 1) * 64, l15 = lane & 15, l4 = lane >> 4;
  for (int i = threadIdx.x; i < 2 * 16 * LDP; i += 256) {
    (&la[0][0][0])[i] = seed * (i & 7);
    (&lb[0][0][0])[i] = seed + (i & 3);
  }
  __syncthreads();
  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = seed + i + lane;
    b[i] = seed - i;
  }
  const int t = threadIdx.x;
  size_t pos = ((size_t)blockIdx.x * 8191 * 2048) % (src_elems - 4096 * 64);
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    dvec2 g[8];
    if (MODE >= 4) {
      // A-like: thread = row, 64 contiguous bytes; B-like: 256 contiguous bytes per 16 threads
      const dvec2* pa = reinterpret_cast<const dvec2*>(src + pos + (size_t)(t & 127) * 512 + (t >> 7) * 8);
      const double* pb = src + pos + 65536 + (size_t)(t >> 4) * 4096 + (t & 15) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[i] = pa[i];
        g[4 + i] = *reinterpret_cast<const dvec2*>(pb + 32 * i);
      }
      pos += 16;
    }
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = la[buf][ks + l4][wr + 16 * i + l15];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = lb[buf][ks + l4][wc + 16 * j + l15];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      if (MODE >= 3 && ks == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dvec2 va = MODE >= 4 ? g[i] : (dvec2){seed * it, seed + i};
          la[buf ^ 1][(t >> 7) * 8 + 2 * i][t & 127] = va.x;
          la[buf ^ 1][(t >> 7) * 8 + 2 * i + 1][t & 127] = va.y;
          dvec2 vb = MODE >= 4 ? g[4 + i] : (dvec2){seed - it, seed * i};
          *reinterpret_cast<dvec2*>(&lb[buf ^ 1][t >> 4][32 * i + 2 * (t & 15)]) = vb;
        }
      }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 5 && (it & 31) == 31) {
      // tile boundary: results out (row-per-lane stores of the MFMA layout), accumulators cleared, first panel in
      double* y = dst + ((size_t)blockIdx.x * 128 * 128 + (size_t)(it >> 5) * 7 * 128 * 128) % (src_elems - 128 * 4096);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int v = 0; v < 4; ++v) y[(size_t)(wr + 16 * i + l4 + 4 * v) * 4096 + wc + 16 * j + l15] = acc[i][j][v];
          acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
        }
      const dvec2* pa = reinterpret_cast<const dvec2*>(src + pos + (size_t)(t & 127) * 512 + (t >> 7) * 8);
      dvec2 va = pa[0];
      la[buf ^ 1][(t >> 7) * 8][t & 127] = va.x + va.y;
      __syncthreads();
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(double* out, int wgs, int iters, const double* src, double* dst, size_t n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_rate<MODE><<<wgs, 256>>>(out, 64, 1e-3, src, dst, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_rate<MODE><<<wgs, 256>>>(out, iters, 1e-3, src, dst, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)wgs * 4 * iters * 64 * 2048.0;
  printf("{\"mode\": %d, \"workgroups\": %d, \"iters\": %d, \"ms\": %.3f, \"tflops\": %.2f}\n", MODE, wgs, iters, ms,
         flop / ms / 1e9);
}

int main(int argc, char** argv) {
  int per_cu = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 2000;
  int wgs = 256 * per_cu;
  double* out;
  hipMalloc(&out, (size_t)wgs * 256 * 8);
  size_t n = (size_t)1 << 27;  // 1 GiB of doubles each
  double *src, *dst;
  hipMalloc(&src, n * 8);
  hipMalloc(&dst, n * 8);
  hipMemset(src, 0, n * 8);
  run<0>(out, wgs, iters, src, dst, n);
  run<1>(out, wgs, iters, src, dst, n);
  run<2>(out, wgs, iters, src, dst, n);
  run<3>(out, wgs, iters, src, dst, n);
  run<4>(out, wgs, iters, src, dst, n);
  run<5>(out, wgs, iters, src, dst, n);
  return 0;
}
