// What fp64 vector rate does an MI355X sustain without FMA?  8 independent mul/add chains per lane (the shape
// of k_traj_gauss_q's inner loop: 40 fp64 instructions per step for 8 rows), nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/kbench/bin/fp64_rate_bench tools/kbench/fp64_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int ROWS>
__global__ __launch_bounds__(256) void k_rate(double* out, double eps, double lam, int steps) {
  double th[ROWS], r[ROWS], t[ROWS];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    th[k] = 1e-3 * (i + k);
    r[k] = 1e-4 * (i - k);
    t[k] = -(lam * th[k]);
  }
  for (int n = 0; n < steps; ++n) {
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      r[k] = r[k] + eps * t[k];
      th[k] = th[k] + eps * r[k];
      t[k] = -(lam * th[k]);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < ROWS; ++k) s += th[k] + r[k];
  out[i] = s;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 4096;
  const int wg_per_cu = argc > 2 ? atoi(argv[2]) : 16;  // x 4 wavefronts = wavefronts per CU (4 SIMDs)
  const int blocks = 256 * wg_per_cu, threads = 256;
  double* out;
  (void)hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_rate<8><<<blocks, threads>>>(out, 0.006, 1.5, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * 8 * (double)steps * blocks * threads;
    printf("%d wavefronts per SIMD, 8 rows x %d steps, %d lanes: %.3f ms, %.2f TFLOP/s fp64 (mul and add counted as 1 each; no-FMA ceiling 39.3 at 2.4 GHz)\n",
           wg_per_cu, steps, blocks * threads, ms, flop / ms / 1e9);
  }
  return 0;
}
