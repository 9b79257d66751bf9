// Example USER target for bayes_kit_amd.CTarget: a model compiled into its own shared library
// and plugged in below the samplers through the one-function plugin ABI of include/bkhip.h
// (bk_target_fn) -- the C form of GradModel.log_density_gradient (bayes_kit/typing.py:25-27)
// for all chains at once.
//
// Density: a stationary-free AR(1) chain over the coordinates,
//     r_0 = theta_0,  r_d = theta_d - a*theta_{d-1}   (d >= 1)
//     logp  = -0.5/s2 * sum_d r_d^2
//     grad_d = -(r_d - a*r_{d+1})/s2                  (r_D := 0)
// Unlike the library's built-in targets its gradient couples neighbouring coordinates.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -ffp-contract=off ar1_target.hip -o libar1_target.so
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Ar1Params {  // what `params` points to (host memory; read at launch time)
  double a;
  double s2;
};

// one lane per chain, coordinates walked in order: loads and stores are coalesced across chains
__global__ __launch_bounds__(256) void k_ar1(const double* theta, double* grad, double* logp, int64_t ld, double a,
                                             double inv_s2, int64_t C, int64_t D) {
  int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double prev = 0.0, r = 0.0, ss = 0.0;
  for (int64_t d = 0; d < D; ++d) {
    double t = theta[d * ld + c];
    double rn = t - a * prev;  // r_d
    if (d > 0 && grad) grad[(d - 1) * ld + c] = -((r - a * rn) * inv_s2);
    ss = ss + rn * rn;
    r = rn;
    prev = t;
  }
  if (D > 0 && grad) grad[(D - 1) * ld + c] = -(r * inv_s2);
  if (logp) logp[c] = (-0.5 * inv_s2) * ss;
}

extern "C" int ar1_target(const double* theta, double* grad, double* logp, int64_t ld, const void* params, int64_t C,
                          int64_t D, void* stream) {
  if (!theta || !params || C < 0 || D < 0 || ld < C) return -1;
  if (C == 0) return 0;
  const Ar1Params* p = static_cast<const Ar1Params*>(params);
  k_ar1<<<dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      theta, grad, logp, ld, p->a, 1.0 / p->s2, C, D);
  return (int)hipGetLastError();
}
