// Example USER target with BOTH forms of the plugin ABI of include/bkhip.h:
//
//   funnel_target    (bk_target_fn)    chain count from the host: every sampler can use it
//   funnel_target_n  (bk_target_fn_n)  chain count read from DEVICE memory: DrGhmcDiag then keeps the sizes of its
//                                      data-dependent lane sets on the device and replays a whole delayed-rejection
//                                      draw -- one call of this function per leapfrog step, drghmc.py:280-283 -- as
//                                      one hipGraph, without a host synchronisation
//
//   bk.CTarget("libfunnel_target.so", "funnel_target", D, counted_symbol="funnel_target_n")
//
// Density: Neal's funnel, v = theta_0 ~ N(0, 9), theta_d ~ N(0, e^v) for d >= 1:
//     s     = sum_{d>=1} theta_d^2,  ev = exp(-v),  n = D - 1
//     logp  = ((-(v*v)/18) - (0.5*n)*v) - (0.5*ev)*s
//     grad0 = ((-v/9) - 0.5*n) + (0.5*ev)*s ;  grad_d = -(ev*theta_d)
// The sum s is formed in the order the library documents for its own funnel (include/bkhip.h, "funnel"; 16
// interleaved class sums -> 4 group sums -> total), so that this plugin and bk.Funnel give the same bits; any
// fixed order would be a valid target.  No params.
//
// Geometry: lane = chain (coalesced rows), a workgroup of 4 wavefronts serves 64 chains, wavefront w sums the
// classes w, w+4, w+8, w+12 and the four group sums meet in LDS.  A workgroup whose chains lie past the count
// exits after one scalar load: that is all the counted form asks of a target.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -ffp-contract=off funnel_target.hip -o libfunnel_target.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkhip_math.h"  // bk_exp: the library's exp (the built-in funnel uses it; same double, same draws)

namespace {

constexpr int WAVES = 4;

// SLOTS > 0: rows per class held in registers (D - 1 <= 16 * SLOTS): one batch of loads, no second pass over
// theta -- on a small lane set a launch is a chain of memory round trips, so that is what its time is made of.
// SLOTS = 0: any D, two passes.
template <int SLOTS>
__global__ __launch_bounds__(64 * WAVES) void k_funnel_plugin(const double* theta, double* grad, double* logp,
                                                              int64_t ld, int64_t C, int64_t D,
                                                              const uint32_t* n_dev) {
  __shared__ double q[WAVES][64];
  int64_t n = C;
  if (n_dev) {  // the counted form: the number of chains is whatever an earlier launch left there
    const int64_t m = (int64_t)*n_dev;
    n = m < n ? m : n;
  }
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  if (c0 >= n) return;  // whole workgroup past the set (uniform: before the barrier)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t c = c0 + lane;
  const bool on = c < n;
  // group w: classes w, w+4, w+8, w+12; class k holds rows 1 + k, 1 + k + 16, ... in order
  double x[SLOTS > 0 ? 4 * SLOTS : 1];
  double cs[4];
  if (SLOTS > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int64_t d = 1 + w + 4 * k + 16 * i;
        x[k * SLOTS + i] = (on && d < D) ? theta[d * ld + c] : 0.0;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i)
        if (1 + w + 4 * k + 16 * i < D) acc = acc + x[k * SLOTS + i] * x[k * SLOTS + i];
      cs[k] = acc;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double acc = 0.0;
      for (int64_t d = 1 + w + 4 * k; d < D; d += 16) {
        const double t = on ? theta[d * ld + c] : 0.0;
        acc = acc + t * t;
      }
      cs[k] = acc;
    }
  }
  const double v = on ? theta[c] : 0.0;
  q[w][lane] = ((cs[0] + cs[1]) + cs[2]) + cs[3];
  __syncthreads();
  if (!on) return;
  const double s = ((q[0][lane] + q[1][lane]) + q[2][lane]) + q[3][lane];
  const double ev = bk_exp(-v);
  const double hn = 0.5 * (double)(D - 1);
  const double he = 0.5 * ev;
  if (w == 0) {
    if (logp) logp[c] = ((-(v * v) / 18.0) - hn * v) - he * s;
    if (grad) grad[c] = ((-v / 9.0) - hn) + he * s;
  }
  if (!grad) return;
  if (SLOTS > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int64_t d = 1 + w + 4 * k + 16 * i;
        if (d < D) grad[d * ld + c] = -(ev * x[k * SLOTS + i]);
      }
  } else {
    for (int64_t d = 1 + w; d < D; d += 4) grad[d * ld + c] = -(ev * theta[d * ld + c]);
  }
}

int launch(const double* theta, double* grad, double* logp, int64_t ld, int64_t C, int64_t D, const uint32_t* n_dev,
           void* stream) {
  if (!theta || (!grad && !logp) || C < 0 || D < 1 || ld < C) return -1;
  if (C == 0) return 0;
  const dim3 grid((unsigned)((C + 63) / 64)), block(64 * WAVES);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (D - 1 <= 32) k_funnel_plugin<2><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
  else if (D - 1 <= 64) k_funnel_plugin<4><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
  else if (D - 1 <= 128) k_funnel_plugin<8><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
  else k_funnel_plugin<0><<<grid, block, 0, s>>>(theta, grad, logp, ld, C, D, n_dev);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int funnel_target(const double* theta, double* grad, double* logp, int64_t ld, const void* /*params*/,
                             int64_t C, int64_t D, void* stream) {
  return launch(theta, grad, logp, ld, C, D, nullptr, stream);
}

extern "C" int funnel_target_n(const double* theta, double* grad, double* logp, int64_t ld, const void* /*params*/,
                               int64_t C, int64_t D, const uint32_t* n_dev, void* stream) {
  if (!n_dev) return -1;
  return launch(theta, grad, logp, ld, C, D, n_dev, stream);
}
