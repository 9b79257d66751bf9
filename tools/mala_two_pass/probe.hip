// MALA step as TWO streaming passes (route b) -- an EXPERIMENT, not part of the library:
//   pass A  k_sums_mask      : read theta, grad, theta', grad' (32 D), the two proposal densities' sums, the decision -> mask, lp
//   pass B  k_select_propose : per element read only the WINNER (theta' / grad' where accepted, theta / grad where not; both
//                              where the two chains of a 16-byte lane disagree), rewrite theta / grad in place where accepted,
//                              read the next draw's normals and write the next proposal
// Optimistic on purpose (normals already in the state layout; no stream bookkeeping): if this does not beat k_mala_step with the
// generator beside it, the productised version will not either.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int64_t i64;
typedef double dvec2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_sums_mask(const double* th, const double* g, const double* thp, const double* gp, i64 ld,
                                                   double* lp, const double* lp_p, const double* log_u, double eps, uint8_t* mask,
                                                   i64 C2, i64 D) {
  __shared__ dvec2 part[2][4][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const i64 c2 = (i64)blockIdx.x * 64 + lane;
  const i64 Dq = (D + 3) / 4, dlo = w * Dq, dhi = dlo + Dq < D ? dlo + Dq : D;
  dvec2 sf = {0.0, 0.0}, sr = {0.0, 0.0};
  if (c2 < C2) {
    for (i64 d0 = dlo; d0 < dhi; d0 += 4) {
      dvec2 a[4], b[4], p[4], q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (d0 + u < dhi) {
          const i64 o = (d0 + u) * ld + 2 * c2;
          a[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(th + o));
          b[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(g + o));
          p[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(thp + o));
          q[u] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(gp + o));
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (d0 + u < dhi) {
          const double xf0 = (p[u].x - a[u].x) - eps * b[u].x, xf1 = (p[u].y - a[u].y) - eps * b[u].y;
          const double xr0 = (a[u].x - p[u].x) - eps * q[u].x, xr1 = (a[u].y - p[u].y) - eps * q[u].y;
          sf.x = sf.x + xf0 * xf0; sf.y = sf.y + xf1 * xf1;
          sr.x = sr.x + xr0 * xr0; sr.y = sr.y + xr1 * xr1;
        }
    }
  }
  part[0][w][lane] = sf;
  part[1][w][lane] = sr;
  __syncthreads();
  if (w == 0 && c2 < C2) {
    dvec2 tf = part[0][0][lane], tr = part[1][0][lane];
    for (int k = 1; k < 4; ++k) { tf.x += part[0][k][lane].x; tf.y += part[0][k][lane].y; tr.x += part[1][k][lane].x; tr.y += part[1][k][lane].y; }
    const double kk = -0.25 / eps;
    const i64 c = 2 * c2;
    const bool a0 = log_u[c] < (lp_p[c] - lp[c]) + (kk * tr.x - kk * tf.x);
    const bool a1 = log_u[c + 1] < (lp_p[c + 1] - lp[c + 1]) + (kk * tr.y - kk * tf.y);
    mask[c] = a0; mask[c + 1] = a1;
    if (a0) lp[c] = lp_p[c];
    if (a1) lp[c + 1] = lp_p[c + 1];
  }
}

template <int ROWS>
__global__ __launch_bounds__(256) void k_select_propose(double* th, double* g, double* thp, const double* gp, const double* z, i64 ld,
                                                        const uint8_t* mask, double eps, double s, i64 C2, i64 D) {
  const i64 c2 = (i64)blockIdx.x * 256 + threadIdx.x, d0 = (i64)blockIdx.y * ROWS;
  if (c2 >= C2) return;
  const bool m0 = mask[2 * c2], m1 = mask[2 * c2 + 1];
  const bool any = m0 | m1, all = m0 & m1;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    if (d0 + i >= D) break;
    const i64 o = (d0 + i) * ld + 2 * c2;
    dvec2 a = {0, 0}, b = {0, 0}, p = {0, 0}, q = {0, 0};
    if (!all) { a = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(th + o)); b = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(g + o)); }
    if (any) { p = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(thp + o)); q = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(gp + o)); }
    const dvec2 zz = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(z + o));
    dvec2 tn, gn;
    tn.x = m0 ? p.x : a.x; tn.y = m1 ? p.y : a.y;
    gn.x = m0 ? q.x : b.x; gn.y = m1 ? q.y : b.y;
    if (any) {
      __builtin_nontemporal_store(tn, reinterpret_cast<dvec2*>(th + o));
      __builtin_nontemporal_store(gn, reinterpret_cast<dvec2*>(g + o));
    }
    dvec2 pn;
    pn.x = (tn.x + eps * gn.x) + s * zz.x;
    pn.y = (tn.y + eps * gn.y) + s * zz.y;
    __builtin_nontemporal_store(pn, reinterpret_cast<dvec2*>(thp + o));
  }
}

extern "C" int probe_sums_mask(const double* th, const double* g, const double* thp, const double* gp, i64 ld, double* lp,
                               const double* lp_p, const double* log_u, double eps, uint8_t* mask, i64 C, i64 D, void* stream) {
  k_sums_mask<<<dim3((unsigned)((C / 2 + 63) / 64)), dim3(256), 0, (hipStream_t)stream>>>(th, g, thp, gp, ld, lp, lp_p, log_u, eps, mask, C / 2, D);
  return (int)hipGetLastError();
}
extern "C" int probe_select_propose(double* th, double* g, double* thp, const double* gp, const double* z, i64 ld, const uint8_t* mask,
                                    double eps, double s, i64 C, i64 D, void* stream) {
  k_select_propose<2><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)((D + 1) / 2)), dim3(256), 0, (hipStream_t)stream>>>(
      th, g, thp, gp, z, ld, mask, eps, s, C / 2, D);
  return (int)hipGetLastError();
}
