// Experiment: MALA step as a cooperative D-split kernel (32 chains x 256 dims per workgroup, two
// workgroups per CU, per-chain partial sums exchanged through global memory).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int64_t i64;
typedef double dvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int bk_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)); }
template <bool NT> __device__ __forceinline__ dvec2 ld2(const double* p) {
  const dvec2* q = reinterpret_cast<const dvec2*>(p);
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT> __device__ __forceinline__ void st2(double* p, dvec2 v) {
  dvec2* q = reinterpret_cast<dvec2*>(p);
  if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}

#define BO(arr, o) reinterpret_cast<const double*>(reinterpret_cast<const char*>(arr) + (o))
#define BOW(arr, o) reinterpret_cast<double*>(reinterpret_cast<char*>(arr) + (o))
#ifndef COOP_T
#define COOP_T 512
#endif
constexpr int T = COOP_T, P = 16, CH = 32, ROWS = T / P, WAVES = T / 64;
#ifndef COOP_SLAB
#define COOP_SLAB 256
#endif
#ifndef COOP_WPS
#define COOP_WPS (COOP_T == 512 ? 4 : 2)
#endif
constexpr int EE = COOP_SLAB / ROWS;  // slots per thread
// E slots: slab of ROWS*E dims
template <int E, bool NT, bool XR>
__global__ __launch_bounds__(T, COOP_WPS) void k_coop(const double* th, double* out, double* g, double* thp, const double* gp,
                                               i64 ld, double* lp, const double* lp_p, const double* log_u,
                                               const double* zt, i64 ldz, double eps, double s, double* part,
                                               unsigned* cnt, unsigned* err, int S, i64 C, i64 D) {
  constexpr int SLAB = ROWS * E, ZPITCH = SLAB + 2;
  constexpr int QB = E * T * 16, ZB = CH * ZPITCH * 8, BIG = QB > ZB ? QB : ZB;
  __shared__ __attribute__((aligned(16))) unsigned char big[BIG];
  __shared__ double red[WAVES * P * 4];
  dvec2* qs = reinterpret_cast<dvec2*>(big);
  double* zs = reinterpret_cast<double*>(big);
  const int t = threadIdx.x, j = t % P, r = t / P, lane = t & 63, w = bk_wave_id();
  // block -> (chain block, slab)
  i64 cbk; int sl;
  {
    const i64 bid = blockIdx.x, nCB = gridDim.x / S;
    if (XR && (8 % S == 0) && (nCB % (8 / S) == 0)) {
      const int x = (int)(bid % 8); const i64 q = bid / 8;
      sl = x % S; cbk = (i64)(x / S) * (nCB / (8 / S)) + q;
    } else { cbk = bid / S; sl = (int)(bid % S); }
  }
  const i64 cb = cbk * CH, c = cb + 2 * j;
  const bool cok = c < C;
  const int d0 = sl * SLAB;
  // Loads are UNCONDITIONAL (straight-line code: no per-slot branches around 16-register tuples):
  // out-of-range rows / chain pairs read a clamped, valid address and their values are never used.
  const i64 cl_ = cok ? c : (C - 2);
  dvec2 a[E], b[E], p[E];
  unsigned okm = 0;   // bit e: slot e holds a real (row, chain pair)
  unsigned off[E];    // byte offsets (arrays < 4 GiB)
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const i64 d = d0 + r + ROWS * e;
    if (cok && d < D) okm |= 1u << e;
    off[e] = (unsigned)(((d < D ? d : D - 1) * ld + cl_) * 8);  // BYTE offset: uniform base + u32 offset addressing
  }
#pragma unroll
  for (int e = 0; e < E; ++e)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(gp) + off[e]),
                                     (__attribute__((address_space(3))) void*)(big + (e * T + w * 64) * 16), 16, 0, NT ? 2 : 0);
#pragma unroll
  for (int e = 0; e < E; ++e) { a[e] = ld2<NT>(BO(th, off[e])); b[e] = ld2<NT>(BO(g, off[e])); p[e] = ld2<NT>(BO(thp, off[e])); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double sf0 = 0, sf1 = 0, sr0 = 0, sr1 = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * T + t];
    const double xf0 = (p[e].x - a[e].x) - eps * b[e].x, xf1 = (p[e].y - a[e].y) - eps * b[e].y;
    const double xr0 = (a[e].x - p[e].x) - eps * q.x, xr1 = (a[e].y - p[e].y) - eps * q.y;
    const double m = ((okm >> e) & 1u) ? 1.0 : 0.0;  // x*x*1.0 == x*x exactly; masked slots add +0.0
    sf0 += m * (xf0 * xf0); sf1 += m * (xf1 * xf1); sr0 += m * (xr0 * xr0); sr1 += m * (xr1 * xr1);
  }
#pragma unroll
  for (int m = P; m < 64; m <<= 1) {
    sf0 += __shfl_xor(sf0, m); sf1 += __shfl_xor(sf1, m); sr0 += __shfl_xor(sr0, m); sr1 += __shfl_xor(sr1, m);
  }
  if (lane < P) { double* o = red + (w * P + j) * 4; o[0] = sf0; o[1] = sf1; o[2] = sr0; o[3] = sr1; }
  __syncthreads();
  double tf0 = 0, tf1 = 0, tr0 = 0, tr1 = 0;
#pragma unroll
  for (int k = 0; k < WAVES; ++k) { const double* o = red + (k * P + j) * 4; tf0 += o[0]; tf1 += o[1]; tr0 += o[2]; tr1 += o[3]; }
  __shared__ unsigned char dec[CH];
  if (w == 0) {
    // wavefront 0, lanes < P: totals of this slab for chain pair `lane` -> exchange with the other
    // slabs of the chain block through global memory, decide, hand the decisions to the workgroup
    if (S > 1) {
      double* mine = part + ((cbk * S + sl) * P + j) * 4;
      if (lane < P) {
        __hip_atomic_store(mine + 0, tf0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, tf1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, tr0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 3, tr1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        __hip_atomic_fetch_add(cnt + cbk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt + cbk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
          __builtin_amdgcn_s_sleep(32);
          if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      tf0 = tf1 = tr0 = tr1 = 0.0;
      if (lane < P)
        for (int k = 0; k < S; ++k) {
          const double* o = part + ((cbk * S + k) * P + j) * 4;
          tf0 += __hip_atomic_load(o + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tf1 += __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tr0 += __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tr1 += __hip_atomic_load(o + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane < P) {
      bool d0_ = false, d1_ = false;
      if (cok) {
        const double k = -0.25 / eps;
        d0_ = log_u[c] < (lp_p[c] - lp[c]) + (k * tr0 - k * tf0);
        d1_ = log_u[c + 1] < (lp_p[c + 1] - lp[c + 1]) + (k * tr1 - k * tf1);
      }
      dec[2 * j] = d0_; dec[2 * j + 1] = d1_;
    }
  }
  __syncthreads();
  const bool acc0 = dec[2 * j] != 0, acc1 = dec[2 * j + 1] != 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * T + t];
    a[e].x = acc0 ? p[e].x : a[e].x; a[e].y = acc1 ? p[e].y : a[e].y;
    b[e].x = acc0 ? q.x : b[e].x; b[e].y = acc1 ? q.y : b[e].y;
    if ((okm >> e) & 1u) { st2<NT>(BOW(out, off[e]), a[e]); st2<NT>(BOW(g, off[e]), b[e]); }
  }
  if (!zt) return;
  __syncthreads();
  // stage the normals of this slab: 32 chains x SLAB dims, chain-major rows of SLAB*8 bytes
#pragma unroll
  for (int h = 0; h < CH / WAVES; ++h) {
    const int cl = (CH / WAVES) * w + h;
    const i64 cc = cb + cl;
#pragma unroll
    for (int k = 0; k < (SLAB + 127) / 128; ++k) {
      const int d = 2 * lane + 128 * k;
      if (cc < C && d < SLAB && d0 + d < D)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zt + cc * ldz + d0 + d),
                                         (__attribute__((address_space(3))) void*)(big + (cl * ZPITCH + 128 * k) * 8), 16, 0, NT ? 2 : 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int d = r + ROWS * e;
    const double z0 = zs[(2 * j) * ZPITCH + d], z1 = zs[(2 * j + 1) * ZPITCH + d];
    dvec2 pn; pn.x = (a[e].x + eps * b[e].x) + s * z0; pn.y = (a[e].y + eps * b[e].y) + s * z1;
    if ((okm >> e) & 1u) st2<NT>(BOW(thp, off[e]), pn);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const i64 C = argc > 1 ? atoll(argv[1]) : 65536, D = argc > 2 ? atoll(argv[2]) : 1024;
  const i64 n = C * D; const i64 ldz = (D + 7) / 8 * 8;
  double *th, *out, *g, *thp, *gp, *zt, *lp, *lpp, *lu, *part; unsigned *cnt, *err;
  CK(hipMalloc(&th, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&g, n * 8)); CK(hipMalloc(&thp, n * 8));
  CK(hipMalloc(&gp, n * 8)); CK(hipMalloc(&zt, C * ldz * 8)); CK(hipMalloc(&lp, C * 8)); CK(hipMalloc(&lpp, C * 8)); CK(hipMalloc(&lu, C * 8));
  const int E = EE, SLAB = ROWS * E; const int S = (int)((D + SLAB - 1) / SLAB); const i64 nCB = (C + CH - 1) / CH;
  CK(hipMalloc(&part, nCB * S * P * 4 * 8)); CK(hipMalloc(&cnt, nCB * 4 + 4)); err = cnt + nCB;
  std::vector<double> h(n);
  for (i64 i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
  for (double* p : {th, out, g, thp, gp}) CK(hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(zt, h.data(), C * ldz * 8, hipMemcpyHostToDevice));
  for (double* p : {lp, lpp, lu}) CK(hipMemcpy(p, h.data(), C * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-44s %8.1f us  %7.2f TB/s  err=%u\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12, herr);
    CK(hipGetLastError());
  };
  const double B = (double)n * 8;
  printf("C=%lld D=%lld S=%d nCB=%lld grid=%lld\n", (long long)C, (long long)D, S, (long long)nCB, (long long)(nCB * S));
  timeit("coop E=8 NT XR", 8 * B, [&] { CK(hipMemsetAsync(cnt, 0, nCB * 4 + 4)); k_coop<EE, true, true><<<dim3((unsigned)(nCB * S)), dim3(T)>>>(th, out, g, thp, gp, C, lp, lpp, lu, zt, ldz, 1e-4, 0.0141, part, cnt, err, S, C, D); });
  timeit("coop E=8 NT noXR", 8 * B, [&] { CK(hipMemsetAsync(cnt, 0, nCB * 4 + 4)); k_coop<EE, true, false><<<dim3((unsigned)(nCB * S)), dim3(T)>>>(th, out, g, thp, gp, C, lp, lpp, lu, zt, ldz, 1e-4, 0.0141, part, cnt, err, S, C, D); });
  timeit("coop E=8 plain XR", 8 * B, [&] { CK(hipMemsetAsync(cnt, 0, nCB * 4 + 4)); k_coop<EE, false, true><<<dim3((unsigned)(nCB * S)), dim3(T)>>>(th, out, g, thp, gp, C, lp, lpp, lu, zt, ldz, 1e-4, 0.0141, part, cnt, err, S, C, D); });
  timeit("coop E=8 NT XR no-z (6 arrays)", 6 * B, [&] { CK(hipMemsetAsync(cnt, 0, nCB * 4 + 4)); k_coop<EE, true, true><<<dim3((unsigned)(nCB * S)), dim3(T)>>>(th, out, g, thp, gp, C, lp, lpp, lu, nullptr, ldz, 1e-4, 0.0141, part, cnt, err, S, C, D); });
  return 0;
}
