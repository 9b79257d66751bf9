/* A complete many-chain HMC loop (bayes_kit/hmc.py:55-63 for C chains at once) written against
 * NOTHING but the C ABI of include/bkhip.h and the HIP runtime: no Python, no PyTorch.  This is what a
 * host in any language with a C FFI does; bayes_kit_amd/hmc.py issues the same calls through ctypes.
 *
 *   gcc -std=gnu11 -O2 -D__HIP_PLATFORM_AMD__ examples/c_host/hmc_main.c -I/opt/rocm/include -Iinclude \
 *       -Lbayes-kit_amd/bayes_kit_amd/lib -lbkhip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,'$ORIGIN/../../bayes-kit_amd/bayes_kit_amd/lib' -Wl,-rpath,/opt/rocm/lib -o examples/c_host/hmc_main
 *   ./examples/c_host/hmc_main [chains dims steps draws seed]
 *
 * Target: diagonal Gaussian, lam = linspace(1, 2, D) (bk_target_diag_gaussian_grad); eps = 0.05.
 * Prints the accept rate and, for chain 0 and the last chain, the final theta as hex words, so that
 * tests/test_gpu_samplers.py can compare them bit for bit with the Python driver (and the oracle).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bkhip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    int rc_ = (int)(x);                                                         \
    if (rc_ != 0) {                                                             \
      fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, rc_);        \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

static size_t g_stagger = 0, g_nalloc = 0; /* BK_STAGGER: byte offset added per allocation (layout experiment) */
static char* g_slab = NULL; /* BK_SLAB: carve every array out of ONE allocation (layout experiment) */
static size_t g_slab_used = 0;
static double* dalloc(size_t n) {
  char* p = NULL;
  const size_t bytes = n * sizeof(double);
  if (g_slab) { /* consecutive arrays: 2 MiB-rounded size + BK_STAGGER bytes apart */
    p = g_slab + g_slab_used;
    g_slab_used += ((bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) + g_stagger;
    return (double*)p;
  }
  CK(hipMalloc((void**)&p, bytes + 32 * g_stagger)); /* separate allocations: the n-th is shifted by n * BK_STAGGER */
  return (double*)(p + ((g_nalloc++) & 31) * g_stagger);
}

int main(int argc, char** argv) {
  const int64_t C = argc > 1 ? atoll(argv[1]) : 4096, D = argc > 2 ? atoll(argv[2]) : 64;
  const int64_t L = argc > 3 ? atoll(argv[3]) : 8, draws = argc > 4 ? atoll(argv[4]) : 10;
  const uint64_t seed = argc > 5 ? strtoull(argv[5], NULL, 10) : 2024;
  const double eps = 0.05, half = 0.5 * eps;
  const int64_t ld = C + (getenv("BK_PAD") ? atoll(getenv("BK_PAD")) : 0); /* leading dimension (row pitch) */
  g_stagger = getenv("BK_STAGGER") ? (size_t)atoll(getenv("BK_STAGGER")) : 0;
  if (getenv("BK_SLAB")) CK(hipMalloc((void**)&g_slab, (size_t)(7 * D * ld + 16 * C + D) * sizeof(double) + ((size_t)64 << 20) + 24 * g_stagger));
  hipStream_t s;
  CK(hipStreamCreate(&s));

  /* per-chain Philox streams: key = (seed, chain id), numpy's bit_generator.state layout */
  uint64_t* rng = NULL;
  CK(hipMalloc((void**)&rng, BK_RNG_WORDS * C * sizeof(uint64_t)));
  CK(bk_rng_init_philox(rng, C, seed, 0, C, s));

  double *theta = dalloc(D * ld), *theta_p = dalloc(D * ld), *rho = dalloc(D * ld), *grad = dalloc(D * ld),
         *grad_p = dalloc(D * ld), *lam = dalloc(D), *work = dalloc(bk_refresh_work_elems(C, D));
  double *lp = dalloc(C), *lp_p = dalloc(C), *kin0 = dalloc(C), *kin1 = dalloc(C), *logu = dalloc(C), *ret = dalloc(C);
  uint8_t* mask = NULL;
  uint32_t* accepted = NULL;
  CK(hipMalloc((void**)&mask, C));
  CK(hipMalloc((void**)&accepted, sizeof(uint32_t)));
  CK(hipMemsetAsync(accepted, 0, sizeof(uint32_t), s));
  double* lam_h = (double*)malloc(D * sizeof(double));
  for (int64_t d = 0; d < D; ++d) lam_h[d] = D > 1 ? 1.0 + (double)d / (double)(D - 1) : 1.0;
  CK(hipMemcpyAsync(lam, lam_h, D * sizeof(double), hipMemcpyHostToDevice, s));

  /* theta0 = rng.normal(size=D) from each chain's own stream (hmc.py:24-28) */
  CK(bk_momentum_refresh(BK_RNG_PHILOX, rng, C, NULL, 0.0, 1.0, theta, ld, NULL, NULL, NULL, C, D, work,
                         bk_refresh_work_elems(C, D), s));
  /* (logp, grad) of the current point, kept across draws */
  CK(bk_target_diag_gaussian_grad(theta, grad, lp, ld, lam, C, D, s));

  hipEvent_t ev0, ev1;
  CK(hipEventCreate(&ev0));
  CK(hipEventCreate(&ev1));
  for (int64_t n = 0; n < draws; ++n) {
    if (n == 1) CK(hipEventRecord(ev0, s)); /* timing from the second draw on */
    if (getenv("BK_TRACE") && n % 10 == 1) { /* per-10-draw timing (machine-state drift experiment) */
      static hipEvent_t prev;
      static int have = 0;
      hipEvent_t cur;
      CK(hipEventCreate(&cur));
      CK(hipEventRecord(cur, s));
      if (have) {
        float ms = 0.f;
        CK(hipEventSynchronize(cur));
        CK(hipEventElapsedTime(&ms, prev, cur));
        fprintf(stderr, "draws %lld-%lld: %.3f ms per draw\n", (long long)(n - 10), (long long)(n - 1), ms / 10.0);
      }
      prev = cur;
      have = 1;
    }
    /* rho ~ N(0, I), kin0 = 1/2 rho.rho, then the accept uniform: the reference's stream order */
    CK(bk_momentum_refresh(BK_RNG_PHILOX, rng, C, NULL, 0.0, 1.0, rho, ld, NULL, kin0, NULL, C, D, work,
                           bk_refresh_work_elems(C, D), s));
    CK(bk_log_uniform(BK_RNG_PHILOX, rng, C, logu, NULL, C, s));
    /* leapfrog (hmc.py:40-53): back half step folded into the first kick */
    const double* g = grad;
    for (int64_t k = 0; k < L; ++k) {
      if (k == 0)
        CK(bk_leapfrog_kick_drift(theta, theta_p, rho, rho, ld, g, ld, 1, NULL, eps, 1, -half, 1, eps, C, D, s));
      else
        CK(bk_leapfrog_kick_drift(theta_p, theta_p, rho, rho, ld, g, ld, 1, NULL, eps, 0, 0.0, 1, eps, C, D, s));
      CK(bk_target_diag_gaussian_grad(theta_p, grad_p, k == L - 1 ? lp_p : NULL, ld, lam, C, D, s));
      g = grad_p;
    }
    /* forward half step + kinetic energy of the proposal (hmc.py:52, :37) */
    CK(bk_leapfrog_finish(rho, NULL, ld, g, ld, 1, NULL, half, 0, kin1, C, D, s));
    /* accept iff log(u) < (lp' - kin') - (lp - kin) (hmc.py:57-63); accepted chains take the proposal */
    CK(bk_mh_accept(BK_ACCEPT_HMC, lp, kin0, lp_p, kin1, logu, mask, ret, accepted, C, s));
    CK(bk_select_columns(mask, theta, theta_p, grad, grad_p, NULL, ld, C, D, s));
  }
  CK(hipEventRecord(ev1, s));
  CK(hipStreamSynchronize(s));
  if (draws > 1) {
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, ev0, ev1));
    fprintf(stderr, "%.3f ms per draw, %.4g leapfrog steps/s\n", ms / (double)(draws - 1),
            (double)C * (double)L * (double)(draws - 1) / (ms * 1e-3));
    if (getenv("BK_SHOW_PTRS"))
      fprintf(stderr, "theta %p theta_p %p rho %p grad %p grad_p %p\n", (void*)theta, (void*)theta_p, (void*)rho,
              (void*)grad, (void*)grad_p);
  }

  uint32_t acc = 0;
  CK(hipMemcpy(&acc, accepted, sizeof(acc), hipMemcpyDeviceToHost));
  printf("chains %lld dims %lld steps %lld draws %lld accept %.6f\n", (long long)C, (long long)D, (long long)L,
         (long long)draws, (double)acc / ((double)C * (double)draws));
  double* col = (double*)malloc(D * sizeof(double));
  const int64_t show[2] = {0, C - 1};
  for (int i = 0; i < 2; ++i) {
    CK(hipMemcpy2D(col, sizeof(double), theta + show[i], ld * sizeof(double), sizeof(double), D, hipMemcpyDeviceToHost));
    printf("theta[%lld]", (long long)show[i]);
    for (int64_t d = 0; d < D; ++d) {
      uint64_t w;
      memcpy(&w, &col[d], 8);
      printf(" %016llx", (unsigned long long)w);
    }
    printf("\n");
  }
  return 0;
}
