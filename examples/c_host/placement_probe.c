/* Experiment: how does the relative placement of the three arrays streamed by bk_leapfrog_kick_drift
 * (theta, rho, grad at equal offsets) change its rate?  One slab; theta at 0, rho and grad at
 * 512 MiB / 1 GiB plus offsets (S1, S2) in bytes: a list of hand-picked pairs, then random ones.
 * gcc, C ABI only (see hmc_main.c). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bkhip.h"

#define CK(x) do { int rc_ = (int)(x); if (rc_) { fprintf(stderr, "%s -> %d\n", #x, rc_); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int64_t C = 65536, D = 1024;
  const size_t A = (size_t)D * C * 8;  /* 512 MiB */
  const int trials = argc > 1 ? atoi(argv[1]) : 40, reps = 20;
  char* slab;
  CK(hipMalloc((void**)&slab, 3 * A + ((size_t)1 << 30)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  CK(hipMemsetAsync(slab, 0, 3 * A + ((size_t)1 << 30), s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t fixed[][2] = {{0, 0}, {4096, 8192}, {1 << 20, 2 << 20}, {0x41000, 0x82000}, {0x0A5A5100, 0x15A5A300},
                             {0x01234500, 0x02B67D00}, {0x035E9300, 0x07F3B100}, {0x00155500, 0x002AAA00}};
  const int nfixed = (int)(sizeof(fixed) / sizeof(fixed[0]));
  srand(12345);
  for (int t = 0; t < nfixed + trials; ++t) {
    size_t s1, s2;
    if (t < nfixed) {
      s1 = fixed[t][0];
      s2 = fixed[t][1];
    } else {
      s1 = ((size_t)rand() % (1 << 21)) * 256;  /* < 512 MiB, 256-byte granules */
      s2 = ((size_t)rand() % (1 << 21)) * 256;
    }
    double* th = (double*)slab;
    double* rho = (double*)(slab + A + s1);
    double* g = (double*)(slab + 2 * A + (s1 > s2 ? s1 : s2) + s2);
    if ((char*)g + A > slab + 3 * A + ((size_t)1 << 30)) continue;
    for (int w = 0; w < 3; ++w) CK(bk_leapfrog_kick_drift(th, th, rho, rho, C, g, C, 1, NULL, 0.01, 0, 0.0, 1, 0.01, C, D, s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(bk_leapfrog_kick_drift(th, th, rho, rho, C, g, C, 1, NULL, 0.01, 0, 0.0, 1, 0.01, C, D, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rho-theta %#12zx  grad-theta %#12zx : %.1f us  %.2f TB/s\n", (size_t)((char*)rho - (char*)th),
           (size_t)((char*)g - (char*)th), 1e3 * ms / reps, 40.0 * D * C / (ms / reps * 1e-3) / 1e12);
  }
  return 0;
}
