// The MALA step kernel (bk_mala.hip: proposal densities + accept + select + the NEXT draw's proposal in one pass over a block of
// 16 chains x all dimensions, bayes_kit/mala.py:41-66) for SEPARABLE densities, with the density's term inlined.
//
// A model-opaque draw moves 88*D bytes per chain: the gradient op reads theta' and writes grad' (16 D), the step kernel reads
// theta, grad, theta', grad' (32 D), writes theta_new and grad_new (16 D) and, with the next draw's normals (8 D written by the
// generator, 8 D read), the next proposal (8 D).  When log p = sum_d term(theta_d) the gradient of a coordinate is a function of
// that coordinate alone: grad and grad' are RECOMPUTED from theta and theta' where the model-opaque kernel loads them, and
// grad_new is never stored.  What is left: the model's log-density launch (8 D read; its per-chain sum keeps the order of the
// target's own op, so lp' is the same number), the step kernel's theta, theta' in (16 D), theta_new, theta'' out (16 D), the
// normals (16 D): 56*D.  Same arithmetic on the same values as the model-opaque pair of launches: bit-identical draws.
//
// TERM is bk_elementwise.hpp's: eval(th, d, params, term&, grad&).  Instantiated for the built-in Gaussians (bk_targets.hip) and
// for CTarget.from_source(form="elementwise") densities (bk_source_kernels.hpp).  Code shape, buffer addressing and the fixed
// summation order (rows of a wavefront by a xor tree, the 8 wavefronts in order: a function of D alone) are k_mala_step's.
#pragma once
#include "bk_common.hpp"

namespace bkm {

typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A workgroup holds PAIRS chain pairs x all dimensions: 64 * PAIRS threads, thread (j, r) = (t % PAIRS, t / PAIRS) the rows
// r, r + 64, ... of pair j.  PAIRS = 8 is k_mala_step's shape (16 chains, 128-byte row segments, the whole register file of a
// CU) and the one that runs.  PAIRS = 4 (8 chains, 64-byte segments, one wavefront per SIMD; BK_MALA_STEP_PAIRS=4) leaves room
// for a wavefront of the normals' generator beside it on every SIMD: measured at 65,536 x 1,024 the two kernels then do overlap,
// and each takes about twice as long (step 514 -> 847 us, generator 454 -> 863 us; alone the narrow step takes 681 us) -- 0.97 ms
// per draw either way (profiles/r5_mala.md).  Same summation order for both shapes.
constexpr int MS_ROWS = 64;  // rows per slot
constexpr unsigned RSRC_FLAGS = 0x00020000u;    // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)

__device__ __forceinline__ dvec2 as_d2(u32x4 v) { return __builtin_bit_cast(dvec2, v); }
__device__ __forceinline__ u32x4 as_u4(dvec2 v) { return __builtin_bit_cast(u32x4, v); }

// E row slots per thread: rows r, r+64, ..., r+64(E-1)  (D <= 64 E)
template <class TERM, int E, bool NT, int PAIRS>
__global__ __launch_bounds__(64 * PAIRS) void k_mala_step_sep(const double* th, double* out, double* thp, i64 ld,
                                                              const double* params, double* lp,
                                                              const double* __restrict__ lp_p,
                                                              const double* __restrict__ log_u, const double* zt, i64 ldz,
                                                              double eps, double s, uint8_t* mask, double* ret,
                                                              uint32_t* count, i64 C, i64 D) {
  constexpr int MS_THREADS = 64 * PAIRS, MS_PAIRS = PAIRS, MS_CHAINS = 2 * PAIRS;
  constexpr int ZPITCH = MS_ROWS * E + 2;  // doubles; +2: conflict-free transposed ds_read_b64
  __shared__ __attribute__((aligned(16))) unsigned char big[MS_CHAINS * ZPITCH * 8];
  __shared__ double red[8 * MS_PAIRS * 4];  // [group of 8 rows][pair][4 sums]
  double* zs = reinterpret_cast<double*>(big);
  constexpr int AUX = NT ? 2 : 0;

  const int t = threadIdx.x, j = t % MS_PAIRS, r = t / MS_PAIRS;
  const int lane = t & (BK_WAVE - 1), w = bk_wave_id();  // (wavefront w stages chains 2w, 2w+1 of the block in phase 4)
  unsigned bid = blockIdx.x;  // XCD-aware walk over the chain blocks, as k_mala_step (placement only)
  {
    const unsigned per = gridDim.x / 8u;
    if (bid < per * 8u) bid = (bid % 8u) * per + bid / 8u;
  }
  const i64 cb = (i64)bid * MS_CHAINS;
  const i64 c = cb + 2 * j;
  const bool cok = c < C;  // C is even: a pair is inside or outside as a whole
  const unsigned nbytes = (unsigned)(((D - 1) * ld + C) * 8);
  const unsigned voff = (unsigned)(((i64)r * ld + (cok ? c : C - 2)) * 8);
  const unsigned slotb = (unsigned)(MS_ROWS * ld * 8);
  const __amdgpu_buffer_rsrc_t r_th = __builtin_amdgcn_make_buffer_rsrc((void*)th, 0, nbytes, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, nbytes, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_thp = __builtin_amdgcn_make_buffer_rsrc((void*)thp, 0, nbytes, RSRC_FLAGS);

  // ---- phase 1: load the block, proposal densities (mala.py:50-53, 68-79) with both gradients recomputed ----------
  dvec2 a[E], p[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    a[e] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_th, voff, e * slotb, AUX));
    p[e] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_thp, voff, e * slotb, AUX));
  }
  __builtin_amdgcn_sched_barrier(0);
  double sf0 = 0.0, sf1 = 0.0, sr0 = 0.0, sr1 = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const i64 d = r + MS_ROWS * e;
    // rows >= D were loaded as zeros and add +0.0 to the sums, as in k_mala_step; the term is evaluated with the LAST row's
    // parameters there (straight-line code: no branch per slot) and its result dropped
    const bool rok = d < D;
    const i64 dd = rok ? d : D - 1;
    double tm, b0, b1, q0, q1;
    TERM::eval(a[e].x, dd, params, tm, b0);
    TERM::eval(a[e].y, dd, params, tm, b1);
    TERM::eval(p[e].x, dd, params, tm, q0);
    TERM::eval(p[e].y, dd, params, tm, q1);
    // x = (theta' - theta) - eps*grad ; reverse: (theta - theta') - eps*grad'   (mala.py:78)
    const double xf0 = rok ? (p[e].x - a[e].x) - eps * b0 : 0.0;
    const double xf1 = rok ? (p[e].y - a[e].y) - eps * b1 : 0.0;
    const double xr0 = rok ? (a[e].x - p[e].x) - eps * q0 : 0.0;
    const double xr1 = rok ? (a[e].y - p[e].y) - eps * q1 : 0.0;
    sf0 = sf0 + xf0 * xf0;
    sf1 = sf1 + xf1 * xf1;
    sr0 = sr0 + xr0 * xr0;
    sr1 = sr1 + xr1 * xr1;
    // (slot by slot: hoisting every slot's parameter loads and gradients would not fit the 256 registers of a thread)
    if ((e & 1) == 1) asm volatile("" ::: "memory");
  }
  // the fixed order of k_mala_step (a function of D alone): a xor tree over each group of 8 consecutive rows -- one wavefront's
  // rows there -- then the 8 groups in order
#pragma unroll
  for (int m = MS_PAIRS; m < 8 * MS_PAIRS; m <<= 1) {
    sf0 = sf0 + __shfl_xor(sf0, m);
    sf1 = sf1 + __shfl_xor(sf1, m);
    sr0 = sr0 + __shfl_xor(sr0, m);
    sr1 = sr1 + __shfl_xor(sr1, m);
  }
  if ((r & 7) == 0) {
    double* o = red + ((r >> 3) * MS_PAIRS + j) * 4;
    o[0] = sf0;
    o[1] = sf1;
    o[2] = sr0;
    o[3] = sr1;
  }
  __syncthreads();
  double tf0 = red[j * 4 + 0], tf1 = red[j * 4 + 1], tr0 = red[j * 4 + 2], tr1 = red[j * 4 + 3];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const double* o = red + (k * MS_PAIRS + j) * 4;
    tf0 = tf0 + o[0];
    tf1 = tf1 + o[1];
    tr0 = tr0 + o[2];
    tr1 = tr1 + o[3];
  }

  // ---- phase 2: decision (metropolis.py:70-76; strict <) -----------------------------------------------------------
  bool acc0 = false, acc1 = false;
  {
    const double k = -0.25 / eps;  // mala.py:79
    const double f0 = k * tf0, f1 = k * tf1, v0 = k * tr0, v1 = k * tr1;
    const i64 cs = cok ? c : 0;
    acc0 = cok && (log_u[cs] < (lp_p[cs] - lp[cs]) + (v0 - f0));
    acc1 = cok && (log_u[cs + 1] < (lp_p[cs + 1] - lp[cs + 1]) + (v1 - f1));
  }
  __syncthreads();  // lp is rewritten only after every thread has read it
  if (t < MS_PAIRS) {
    if (cok) {
      if (mask) {
        mask[c] = acc0 ? 1 : 0;
        mask[c + 1] = acc1 ? 1 : 0;
      }
      const double r0 = acc0 ? lp_p[c] : lp[c], r1 = acc1 ? lp_p[c + 1] : lp[c + 1];  // MODEL log density (mala.py:62-66)
      lp[c] = r0;
      lp[c + 1] = r1;
      if (ret) {
        ret[c] = r0;
        ret[c + 1] = r1;
      }
    }
    if (count) {
      unsigned n = (acc0 ? 1u : 0u) + (acc1 ? 1u : 0u);
#pragma unroll
      for (int m = 1; m < MS_PAIRS; m <<= 1) n += __shfl_xor(n, m);
      if (t == 0 && n) atomicAdd(count, n);
    }
  }

  // ---- phase 3: new state (mala.py:62-64), every element rewritten -------------------------------------------------
  const unsigned woff = cok ? voff : nbytes;  // a chain pair past C stores nowhere (dropped by the range check)
#pragma unroll
  for (int e = 0; e < E; ++e) {
    a[e].x = acc0 ? p[e].x : a[e].x;
    a[e].y = acc1 ? p[e].y : a[e].y;
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(a[e]), r_out, woff, e * slotb, AUX);
    // (the new state as an opaque value: the compiler would otherwise keep BOTH gradients of phase 1 alive across the
    // decision, to select between them in phase 4 -- 64 registers the thread does not have)
    asm volatile("" : "+v"(a[e].x), "+v"(a[e].y));
  }
  if (!zt) return;  // uniform: no next proposal wanted

  // ---- phase 4: next proposal (mala.py:41-45) with the next draw's normals ------------------------------------------
  // zt is chain-major (zt[c*ldz + d]): wavefront w stages chains 2w, 2w+1 of the block by LDS-DMA, 1 KiB per instruction, and
  // every thread then reads its (chain pair, row) elements transposed
  {
    const unsigned zbytes = (unsigned)(((C - 1) * ldz + D) * 8);
    const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc((void*)zt, 0, zbytes, RSRC_FLAGS);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cw = 2 * w + h;
      const i64 cc = (cb + cw < C) ? cb + cw : C - 1;
      const unsigned zoff = (unsigned)(cc * ldz * 8) + 16u * lane;
#pragma unroll
      for (int k = 0; k < (MS_ROWS * E + 127) / 128; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_z, (__attribute__((address_space(3))) void*)(big + (cw * ZPITCH + 128 * k) * 8),
                                                 16, zoff, 1024 * k, 0, AUX);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const i64 d = r + MS_ROWS * e;
    const i64 dd = d < D ? d : D - 1;  // (rows >= D: their stores are dropped by the buffer's range check)
    const double z0 = zs[(2 * j) * ZPITCH + d], z1 = zs[(2 * j + 1) * ZPITCH + d];
    double tm, b0, b1;  // the gradient at the new state: the value the model-opaque kernel selects (same function, same input)
    TERM::eval(a[e].x, dd, params, tm, b0);
    TERM::eval(a[e].y, dd, params, tm, b1);
    dvec2 pn;
    pn.x = (a[e].x + eps * b0) + s * z0;
    pn.y = (a[e].y + eps * b1) + s * z1;
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(pn), r_thp, woff, e * slotb, AUX);
    if ((e & 3) == 3) asm volatile("" ::: "memory");
  }
}

// Host side: the arguments of bk_mala_step (include/bkhip.h) without the two gradient arrays, plus the density's params.
template <class TERM>
static int mala_step_sep_launch(const double* theta, double* theta_out, double* theta_prop, int64_t ld, const double* params,
                                double* lp, const double* lp_prop, const double* log_u, const double* zt_next, int64_t ldz,
                                double eps, double sqrt2eps, uint8_t* accept_mask, double* ret, uint32_t* accept_count,
                                int64_t C, int64_t D, void* stream) {
  if (!theta || !theta_out || !theta_prop || !lp || !lp_prop || !log_u || C < 0 || D < 0) return BK_E_ARG;
  if (C == 0 || D == 0) return BK_OK;
  // 32-bit byte offsets inside every array, with headroom for the slot offsets (bk_mala_step_supported)
  if (!(D <= 1024 && C % 2 == 0 && ld % 2 == 0 && ld >= C && D * ld < ((int64_t)1 << 28))) return BK_E_ALIGN;
  if (!bk_aligned16(theta) || !bk_aligned16(theta_out) || !bk_aligned16(theta_prop)) return BK_E_ALIGN;
  if (zt_next && (ldz < D || ldz % 2 != 0 || !bk_aligned16(zt_next))) return BK_E_ALIGN;
  hipStream_t s = bk_stream(stream);
  // BK_MALA_STEP_PAIRS=4: the narrow workgroups of the note on PAIRS above (experiments)
  static const int pairs = []() { const char* e = getenv("BK_MALA_STEP_PAIRS"); return (e && atoi(e) == 4) ? 4 : 8; }();
  const bool nt = bk_streams_past_llc(4 * C * D);
#define BKM_LAUNCH2(E, NT, P)                                                                                            \
  k_mala_step_sep<TERM, E, NT, P><<<dim3((unsigned)bk_cdiv(C, 2 * P)), dim3(64 * P), 0, s>>>(                            \
      theta, theta_out, theta_prop, ld, params, lp, lp_prop, log_u, zt_next, ldz, eps, sqrt2eps, accept_mask, ret, accept_count, C, \
      D)
#define BKM_LAUNCH(E)                  \
  do {                                 \
    if (pairs == 8) {                  \
      if (nt) BKM_LAUNCH2(E, true, 8); \
      else BKM_LAUNCH2(E, false, 8);   \
    } else {                           \
      if (nt) BKM_LAUNCH2(E, true, 4); \
      else BKM_LAUNCH2(E, false, 4);   \
    }                                  \
  } while (0)
  if (D <= 128) BKM_LAUNCH(2);
  else if (D <= 256) BKM_LAUNCH(4);
  else if (D <= 512) BKM_LAUNCH(8);
  else BKM_LAUNCH(16);
#undef BKM_LAUNCH
#undef BKM_LAUNCH2
  BK_RETURN_LAUNCH_STATUS();
}

}  // namespace bkm
