// Micro-benchmark harness for k_mala_step variants (experiments; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/kbench/mala_step_bench.hip -o gpurun_out/mala_step_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int64_t i64;
#define BK_WAVE 64
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

__device__ unsigned long long* g_ts = nullptr;
#define TS(k) do { if (g_ts && t == 0) g_ts[(i64)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// MODE bit0: loads+sums, bit1: stores of out/g, bit2: z phase + store thp
template <int THREADS, int PAIRS, int E, bool NT, int MODE, int WPS = 1, bool XR = false>
__global__ __launch_bounds__(THREADS, WPS) void k_step(const double* th, double* out, double* g, double* thp, const double* gp,
                                                  i64 ld, double* lp, const double* lp_p, const double* log_u,
                                                  const double* zt, i64 ldz, double eps, double s, i64 C, i64 D) {
  constexpr int ROWS = THREADS / PAIRS, CHAINS = 2 * PAIRS, WAVES = THREADS / 64;
  constexpr int ZPITCH = ROWS * E + 2;
  constexpr int QB = E * THREADS * 16, ZB = CHAINS * ZPITCH * 8, BIG = QB > ZB ? QB : ZB;
  __shared__ __attribute__((aligned(16))) unsigned char big[BIG];
  __shared__ double red[WAVES * PAIRS * 4];
  dvec2* qs = reinterpret_cast<dvec2*>(big);
  double* zs = reinterpret_cast<double*>(big);
  const int t = threadIdx.x, j = t % PAIRS, r = t / PAIRS, lane = t & 63, w = bk_wave_id();
  i64 bid = blockIdx.x;
  if (XR) {  // XCD x (blocks b % 8 == x) walks a contiguous range of chain blocks
    const i64 nb = gridDim.x, per = nb / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
  }
  const i64 cb = bid * CHAINS, c = cb + 2 * j;
  const bool cok = c < C;
  const i64 cl_ = cok ? c : (C - 2);
  unsigned okm = 0;
  unsigned off[E];  // byte offsets
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const i64 d = r + ROWS * e;
    if (cok && d < D) okm |= 1u << e;
    off[e] = (unsigned)(((d < D ? d : D - 1) * ld + cl_) * 8);
  }
#define BO(arr, o) reinterpret_cast<const double*>(reinterpret_cast<const char*>(arr) + (o))
#define BOW(arr, o) reinterpret_cast<double*>(reinterpret_cast<char*>(arr) + (o))
  TS(0);
  dvec2 a[E], b[E], p[E];
#pragma unroll
  for (int e = 0; e < E; ++e)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)BO(gp, off[e]),
                                     (__attribute__((address_space(3))) void*)(big + (e * THREADS + w * 64) * 16), 16, 0, NT ? 2 : 0);
#pragma unroll
  for (int e = 0; e < E; ++e) { a[e] = ld2<NT>(BO(th, off[e])); b[e] = ld2<NT>(BO(g, off[e])); p[e] = ld2<NT>(BO(thp, off[e])); }
  TS(1);
  double sf0 = 0, sf1 = 0, sr0 = 0, sr1 = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * THREADS + t];
    const double xf0 = (p[e].x - a[e].x) - eps * b[e].x, xf1 = (p[e].y - a[e].y) - eps * b[e].y;
    const double xr0 = (a[e].x - p[e].x) - eps * q.x, xr1 = (a[e].y - p[e].y) - eps * q.y;
    const double m = ((okm >> e) & 1u) ? 1.0 : 0.0;
    sf0 += m * (xf0 * xf0); sf1 += m * (xf1 * xf1); sr0 += m * (xr0 * xr0); sr1 += m * (xr1 * xr1);
  }
#pragma unroll
  for (int m = PAIRS; m < 64; m <<= 1) {
    sf0 += __shfl_xor(sf0, m); sf1 += __shfl_xor(sf1, m); sr0 += __shfl_xor(sr0, m); sr1 += __shfl_xor(sr1, m);
  }
  if (lane < PAIRS) { double* o = red + (w * PAIRS + j) * 4; o[0] = sf0; o[1] = sf1; o[2] = sr0; o[3] = sr1; }
  __syncthreads();
  double tf0 = 0, tf1 = 0, tr0 = 0, tr1 = 0;
#pragma unroll
  for (int k = 0; k < WAVES; ++k) { const double* o = red + (k * PAIRS + j) * 4; tf0 += o[0]; tf1 += o[1]; tr0 += o[2]; tr1 += o[3]; }
  bool acc0 = false, acc1 = false;
  if (cok) {
    const double k = -0.25 / eps;
    acc0 = log_u[c] < (lp_p[c] - lp[c]) + (k * tr0 - k * tf0);
    acc1 = log_u[c + 1] < (lp_p[c + 1] - lp[c + 1]) + (k * tr1 - k * tf1);
  }
  TS(2);
  if (!(MODE & 2)) { if (t < PAIRS && cok) { lp[c] = tf0 + (acc0 ? 1 : 0); lp[c + 1] = tf1 + (acc1 ? 1 : 0); } return; }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * THREADS + t];
    a[e].x = acc0 ? p[e].x : a[e].x; a[e].y = acc1 ? p[e].y : a[e].y;
    b[e].x = acc0 ? q.x : b[e].x; b[e].y = acc1 ? q.y : b[e].y;
    if ((okm >> e) & 1u) { st2<NT>(BOW(out, off[e]), a[e]); st2<NT>(BOW(g, off[e]), b[e]); }
  }
  TS(3);
  if (!(MODE & 4)) return;
  __syncthreads();
  TS(4);
  constexpr int CPW = CHAINS / WAVES;  // chains staged per wavefront
#pragma unroll
  for (int h = 0; h < CPW; ++h) {
    const int cl = CPW * w + h;
    const i64 cc = (cb + cl < C) ? cb + cl : C - 1;
#pragma unroll
    for (int k = 0; k < (ROWS * E + 127) / 128; ++k) {
      const int d = 2 * lane + 128 * k;
      if (d < D)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zt + cc * ldz + d),
                                         (__attribute__((address_space(3))) void*)(big + (cl * ZPITCH + 128 * k) * 8), 16, 0, NT ? 2 : 0);
    }
  }
  __syncthreads();
  TS(5);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int d = r + ROWS * e;
    const double z0 = zs[(2 * j) * ZPITCH + d], z1 = zs[(2 * j + 1) * ZPITCH + d];
    dvec2 pn; pn.x = (a[e].x + eps * b[e].x) + s * z0; pn.y = (a[e].y + eps * b[e].y) + s * z1;
    if ((okm >> e) & 1u) st2<NT>(BOW(thp, off[e]), pn);
  }
  TS(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TS(7);
}

// reference points: plain streaming kernels over the same arrays (2 chains x ROWS rows per thread)
template <bool NT, int NR, int NW>
__global__ __launch_bounds__(256) void k_stream(const double* a0, const double* a1, const double* a2, const double* a3,
                                                const double* a4, double* o0, double* o1, double* o2, i64 ld, i64 C2, i64 D) {
  i64 c2 = (i64)blockIdx.x * 256 + threadIdx.x; i64 d0 = (i64)blockIdx.y * 2;
  if (c2 >= C2) return;
  const double* in[5] = {a0, a1, a2, a3, a4}; double* outp[3] = {o0, o1, o2};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    i64 o = (d0 + i) * ld + 2 * c2; dvec2 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < NR; ++k) { dvec2 v = ld2<NT>(in[k] + o); acc.x += v.x; acc.y += v.y; }
#pragma unroll
    for (int k = 0; k < NW; ++k) st2<NT>(outp[k] + o, acc);
    if (NW == 0 && acc.x == 12345.678) o0[o] = acc.y;  // keeps the loads alive
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const i64 C = argc > 1 ? atoll(argv[1]) : 65536, D = argc > 2 ? atoll(argv[2]) : 1024;
  const i64 n = C * D; const i64 ldz = (D + 7) / 8 * 8;
  double *th, *out, *g, *thp, *gp, *zt, *lp, *lpp, *lu;
  CK(hipMalloc(&th, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&g, n * 8)); CK(hipMalloc(&thp, n * 8));
  CK(hipMalloc(&gp, n * 8)); CK(hipMalloc(&zt, C * ldz * 8)); CK(hipMalloc(&lp, C * 8)); CK(hipMalloc(&lpp, C * 8)); CK(hipMalloc(&lu, C * 8));
  std::vector<double> h(n);
  for (i64 i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
  for (double* p : {th, out, g, thp, gp}) CK(hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(zt, h.data(), C * ldz * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(lp, h.data(), C * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(lpp, h.data(), C * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(lu, h.data(), C * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-44s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    CK(hipGetLastError());
  };
  const double B = (double)n * 8;
#define STEP(TH, PA, EE, NTT, MO, bytes, ...) timeit("step T=" #TH " P=" #PA " E=" #EE " NT=" #NTT " MODE=" #MO " " #__VA_ARGS__, bytes, [&] { \
    k_step<TH, PA, EE, NTT, MO, ##__VA_ARGS__><<<dim3((unsigned)((C + 2 * PA - 1) / (2 * PA))), dim3(TH)>>>(th, out, g, thp, gp, C, lp, lpp, lu, zt, ldz, 1e-4, 0.0141, C, D); })
  if (D <= 1024 && D > 512) {
    STEP(512, 8, 16, true, 1, 4 * B);
    STEP(512, 8, 16, true, 3, 6 * B);
    STEP(512, 8, 16, true, 7, 8 * B);
    STEP(512, 8, 16, false, 7, 8 * B);
    STEP(512, 8, 16, true, 1, 4 * B, 1, true);
    STEP(512, 8, 16, true, 7, 8 * B, 1, true);
    STEP(256, 4, 16, false, 1, 4 * B, 2, true);
    STEP(256, 4, 16, false, 7, 8 * B, 2, true);
  }
  if (D > 512) {
    const i64 nb = (C + 15) / 16;
    unsigned long long* ts; CK(hipMalloc(&ts, nb * 8 * 8)); CK(hipMemset(ts, 0, nb * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
    k_step<512, 8, 16, true, 7><<<dim3((unsigned)nb), dim3(512)>>>(th, out, g, thp, gp, C, lp, lpp, lu, zt, ldz, 1e-4, 0.0141, C, D);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nb * 8); CK(hipMemcpy(h.data(), ts, nb * 8 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0; for (i64 b = 0; b < nb; ++b) { if (h[b * 8] < t0) t0 = h[b * 8]; if (h[b * 8 + 7] > t1) t1 = h[b * 8 + 7]; }
    double ph[7] = {0}; for (i64 b = 0; b < nb; ++b) for (int k = 0; k < 7; ++k) ph[k] += (double)(h[b * 8 + k + 1] - h[b * 8 + k]);
    printf("timestamps (s_memtime ticks): kernel span %llu; mean per-WG phase ticks: load %.0f sums+reduce %.0f blend+stores-issue %.0f barrier %.0f z-stage %.0f propose+stores-issue %.0f drain %.0f; sum %.0f\n",
           t1 - t0, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb, ph[4] / nb, ph[5] / nb, ph[6] / nb, (ph[0]+ph[1]+ph[2]+ph[3]+ph[4]+ph[5]+ph[6]) / nb);
    // start-time spread of the first 8 and some later WGs
    for (i64 b : {(i64)0, (i64)1, (i64)255, (i64)256, (i64)257, (i64)2048, nb - 1}) printf("  WG %lld start %llu end %llu\n", (long long)b, h[b * 8] - t0, h[b * 8 + 7] - t0);
    ts = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
  }
  timeit("stream 1R 1W nt", 2 * B, [&] { k_stream<true, 1, 1><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 0R 1W nt", 1 * B, [&] { k_stream<true, 0, 1><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 0R 3W nt", 3 * B, [&] { k_stream<true, 0, 3><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 0R 3W plain", 3 * B, [&] { k_stream<false, 0, 3><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 4R 0W nt", 4 * B, [&] { k_stream<true, 4, 0><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 4R 2W nt", 6 * B, [&] { k_stream<true, 4, 2><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, zt, out, g, thp, C, C / 2, D); });
  timeit("stream 5R 3W nt", 8 * B, [&] { k_stream<true, 5, 3><<<dim3((unsigned)((C / 2 + 255) / 256), (unsigned)(D / 2)), 256>>>(th, g, thp, gp, th, out, g, thp, C, C / 2, D); });
  return 0;
}
