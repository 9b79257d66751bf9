// MALA draw as TWO passes over HBM (bayes_kit/mala.py:40-66).
//
// A draw needs (1) the proposal, (2) the model's gradient at it, (3) both proposal densities --
// per-chain sums over all D dimensions -- (4) the accept decision and (5) the new state.  Steps
// 3-5 cannot be separate elementwise passes without re-reading everything: the decision of a
// chain depends on ALL its dimensions, and the select needs the same four arrays the densities
// were summed from.  k_mala_step therefore gives a workgroup a block of 16 chains x ALL
// dimensions and keeps the block on chip between the sum and the select:
//
//   pass A (model's gradient op, 16*D bytes per chain):   (lp', grad') = model(theta')
//   pass B (k_mala_step):  read theta, grad, theta', grad'                    (32*D)
//                          sums -> decision -> theta_new, grad_new            (16*D written)
//                          and, with the NEXT draw's normals (8*D read), the next proposal
//                          theta'' = (theta_new + eps*grad_new) + s*z         ( 8*D written)
//
// i.e. 88*D bytes per chain-draw including the normals' round trip through the chain-major scratch
// that the wavefront-per-chain generator fills on a side stream (SURVEY 8d's MALA model is 88*D
// with the normals generated in-register; here the propose pass's re-read of theta and grad is
// what is saved instead).
//
// On-chip budget (D <= 1024): 16 chains x 1024 dims x 4 arrays x 8 B = 512 KiB = the whole
// register file of a CU.  Three arrays (theta, grad, theta') stay in registers (192 VGPRs per
// thread at 512 threads), grad' waits in LDS (128 KiB), which afterwards stages the transposition
// of the next draw's normals.  One workgroup per CU; the 256 CUs run out of phase, so HBM sees
// a steady mix of their load and store bursts.
//
// Memory shape: a workgroup's rows are 16 chains = 128 B wide (one L2 line per row and array);
// a wavefront instruction covers 8 rows x 128 B.
#include "bk_common.hpp"

namespace {

typedef double dvec2 __attribute__((ext_vector_type(2)));

constexpr int MS_THREADS = 512;
constexpr int MS_PAIRS = 8;                       // chain pairs (16 B) per row
constexpr int MS_CHAINS = 2 * MS_PAIRS;           // 16 chains per workgroup
constexpr int MS_ROWS = MS_THREADS / MS_PAIRS;    // 64 rows per slot
constexpr int MS_WAVES = MS_THREADS / BK_WAVE;    // 8

template <int E>
struct MsLds {
  static constexpr int ZPITCH = MS_ROWS * E + 2;  // doubles; +2: conflict-free transposed ds_read_b64
  static constexpr int Q_BYTES = E * MS_THREADS * 16;
  static constexpr int Z_BYTES = MS_CHAINS * ZPITCH * 8;
  static constexpr int BIG_BYTES = Q_BYTES > Z_BYTES ? Q_BYTES : Z_BYTES;
  static constexpr int RED_DOUBLES = MS_WAVES * MS_PAIRS * 4;
};

template <bool NT>
__device__ __forceinline__ dvec2 ld2(const double* p) {
  const dvec2* q = reinterpret_cast<const dvec2*>(p);
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void st2(double* p, dvec2 v) {
  dvec2* q = reinterpret_cast<dvec2*>(p);
  if (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}

// E row slots per thread: rows r, r+64, ..., r+64(E-1)  (D <= 64 E)
//
// Code shape (measured, tools/kbench/mala_step_bench.hip):
// * Every global access is a BUFFER instruction: descriptor (SGPRs) + ONE per-thread 32-bit byte
//   offset + a per-slot scalar offset.  No address lives in a VGPR, so the 192 data registers and
//   little else are live (with flat addressing hipcc precomputed 64-bit addresses per slot and
//   spilled); and the descriptor's range check handles the ragged edge for free: rows >= D are
//   past num_records, so their loads return 0 (adding +0.0 to the sums) and their stores are dropped.
// * The body is straight-line; nothing that consumes a load is scheduled into the issue sequence
//   (sched_barrier): with LDS-DMA in flight the first use of any load result waits for vmcnt(0).
// * On gfx950 stores count in vmcnt as well: any vmcnt(0) inside a store phase (a spill reload, a
//   guarded LDS read) serialises the stores by their full latency -- 990 -> 830 us at 65,536 x 1024.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BK_RSRC_FLAGS = 0x00020000u;  // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)

__device__ __forceinline__ dvec2 as_d2(u32x4 v) { return __builtin_bit_cast(dvec2, v); }
__device__ __forceinline__ u32x4 as_u4(dvec2 v) { return __builtin_bit_cast(u32x4, v); }

template <int E, bool NT>
__global__ __launch_bounds__(MS_THREADS) void k_mala_step(
    const double* th, double* out, double* g, double* thp, const double* gp, i64 ld, double* lp,
    const double* __restrict__ lp_p, const double* __restrict__ log_u, const double* zt, i64 ldz, double eps,
    double s, uint8_t* mask, double* ret, uint32_t* count, i64 C, i64 D) {
  using L = MsLds<E>;
  __shared__ __attribute__((aligned(16))) unsigned char big[L::BIG_BYTES];
  __shared__ double red[L::RED_DOUBLES];
  dvec2* qs = reinterpret_cast<dvec2*>(big);
  double* zs = reinterpret_cast<double*>(big);
  constexpr int AUX = NT ? 2 : 0;

  const int t = threadIdx.x, j = t & (MS_PAIRS - 1), r = t >> 3;
  const int lane = t & (BK_WAVE - 1), w = bk_wave_id();
  // XCD x (workgroups b % 8 == x) walks a contiguous range of chain blocks: neighbouring 128-byte
  // row segments then meet in the same L2 (placement affects speed only, never results)
  unsigned bid = blockIdx.x;
  {
    const unsigned per = gridDim.x / 8u;
    if (bid < per * 8u) bid = (bid % 8u) * per + bid / 8u;
  }
  const i64 cb = (i64)bid * MS_CHAINS;
  const i64 c = cb + 2 * j;
  const bool cok = c < C;  // C is even: a pair is inside or outside as a whole
  // arrays are [D][ld]: (D-1)*ld + C elements; a chain pair past C loads pair C-2 instead (unused)
  const unsigned nbytes = (unsigned)(((D - 1) * ld + C) * 8);
  const unsigned voff = (unsigned)(((i64)r * ld + (cok ? c : C - 2)) * 8);
  const unsigned slotb = (unsigned)(MS_ROWS * ld * 8);
  const __amdgpu_buffer_rsrc_t r_th = __builtin_amdgcn_make_buffer_rsrc((void*)th, 0, nbytes, BK_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, nbytes, BK_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, BK_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_thp = __builtin_amdgcn_make_buffer_rsrc((void*)thp, 0, nbytes, BK_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t r_gp = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, nbytes, BK_RSRC_FLAGS);

  // ---- phase 1: load the block, proposal densities (mala.py:50-53, 68-79) --------------------
  // grad' goes global -> LDS directly (LDS-DMA, 16 B per lane, lane-linear destination: slot
  // e*512 + t is written by lane t's own load), so it costs no staging registers while the other
  // three arrays fill 192 VGPRs; all 4 E loads of a thread are in flight together.
  dvec2 a[E], b[E], p[E];
#pragma unroll
  for (int e = 0; e < E; ++e)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_gp, (__attribute__((address_space(3))) void*)(big + (e * MS_THREADS + w * BK_WAVE) * 16),
                                             16, voff, e * slotb, 0, AUX);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    a[e] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_th, voff, e * slotb, AUX));
    b[e] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_g, voff, e * slotb, AUX));
    p[e] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_thp, voff, e * slotb, AUX));
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's grad' slots have landed in LDS
  double sf0 = 0.0, sf1 = 0.0, sr0 = 0.0, sr1 = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * MS_THREADS + t];
    // x = (theta' - theta) - eps*grad ; reverse: (theta - theta') - eps*grad'   (mala.py:78)
    // (rows >= D were loaded as zeros: x = 0 and the sums receive +0.0)
    const double xf0 = (p[e].x - a[e].x) - eps * b[e].x, xf1 = (p[e].y - a[e].y) - eps * b[e].y;
    const double xr0 = (a[e].x - p[e].x) - eps * q.x, xr1 = (a[e].y - p[e].y) - eps * q.y;
    sf0 = sf0 + xf0 * xf0;
    sf1 = sf1 + xf1 * xf1;
    sr0 = sr0 + xr0 * xr0;
    sr1 = sr1 + xr1 * xr1;
    // at most 4 LDS reads ahead: hoisting all E of them would cost 4 E more registers
    if ((e & 3) == 3) asm volatile("" ::: "memory");
  }
  // rows of a wavefront: lanes j, j+8, ..., j+56 -> fixed xor tree, then the 8 wavefronts in order.
  // The summation order depends on D only (never on C or the grid).
#pragma unroll
  for (int m = MS_PAIRS; m < BK_WAVE; m <<= 1) {
    sf0 = sf0 + __shfl_xor(sf0, m);
    sf1 = sf1 + __shfl_xor(sf1, m);
    sr0 = sr0 + __shfl_xor(sr0, m);
    sr1 = sr1 + __shfl_xor(sr1, m);
  }
  if (lane < MS_PAIRS) {
    double* o = red + (w * MS_PAIRS + j) * 4;
    o[0] = sf0; o[1] = sf1; o[2] = sr0; o[3] = sr1;
  }
  __syncthreads();
  double tf0 = red[j * 4 + 0], tf1 = red[j * 4 + 1], tr0 = red[j * 4 + 2], tr1 = red[j * 4 + 3];
#pragma unroll
  for (int k = 1; k < MS_WAVES; ++k) {
    const double* o = red + (k * MS_PAIRS + j) * 4;
    tf0 = tf0 + o[0]; tf1 = tf1 + o[1]; tr0 = tr0 + o[2]; tr1 = tr1 + o[3];
  }

  // ---- phase 2: decision (metropolis.py:70-76; strict <) ---------------------------------------
  // Every thread of a chain pair evaluates the same expression on the same inputs; the pair's
  // first thread (t < 8) publishes.  lp is rewritten only after every thread has read it.
  bool acc0 = false, acc1 = false;
  {
    const double k = -0.25 / eps;  // mala.py:79
    const double f0 = k * tf0, f1 = k * tf1, v0 = k * tr0, v1 = k * tr1;
    const i64 cs = cok ? c : 0;
    acc0 = cok && (log_u[cs] < (lp_p[cs] - lp[cs]) + (v0 - f0));
    acc1 = cok && (log_u[cs + 1] < (lp_p[cs + 1] - lp[cs + 1]) + (v1 - f1));
  }
  __syncthreads();
  if (t < MS_PAIRS) {
    if (cok) {
      if (mask) { mask[c] = acc0 ? 1 : 0; mask[c + 1] = acc1 ? 1 : 0; }
      const double r0 = acc0 ? lp_p[c] : lp[c], r1 = acc1 ? lp_p[c + 1] : lp[c + 1];  // MODEL log density (mala.py:62-66)
      lp[c] = r0; lp[c + 1] = r1;
      if (ret) { ret[c] = r0; ret[c + 1] = r1; }
    }
    if (count) {
      unsigned n = (acc0 ? 1u : 0u) + (acc1 ? 1u : 0u);
      n += __shfl_xor(n, 1); n += __shfl_xor(n, 2); n += __shfl_xor(n, 4);
      if (t == 0 && n) atomicAdd(count, n);
    }
  }

  // ---- phase 3: new state (mala.py:62-64), every element rewritten (blend, no holes) ----------
  // a chain pair past C stores nowhere: its offset is moved past num_records (dropped by the range check)
  const unsigned woff = cok ? voff : nbytes;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const dvec2 q = qs[e * MS_THREADS + t];
    a[e].x = acc0 ? p[e].x : a[e].x;
    a[e].y = acc1 ? p[e].y : a[e].y;
    b[e].x = acc0 ? q.x : b[e].x;
    b[e].y = acc1 ? q.y : b[e].y;
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(a[e]), r_out, woff, e * slotb, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(b[e]), r_g, woff, e * slotb, AUX);
    if ((e & 3) == 3) asm volatile("" ::: "memory");
  }
  if (!zt) return;  // uniform: no next proposal wanted
  __syncthreads();  // grad' no longer needed in LDS

  // ---- phase 4: next proposal (mala.py:41-45) with the next draw's normals -----------------------
  // zt is chain-major (zt[c*ldz + d]): wavefront w stages chains 2w, 2w+1 of the block by LDS-DMA,
  // 1 KiB per instruction, and every thread then reads its (chain pair, row) elements transposed.
  // (reads past a chain's D normals land in the row padding / the next chain's row, or past
  // num_records for the last chain -> 0; those LDS cells are never used)
  {
    const unsigned zbytes = (unsigned)(((C - 1) * ldz + D) * 8);
    const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc((void*)zt, 0, zbytes, BK_RSRC_FLAGS);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cw = 2 * w + h;
      const i64 cc = (cb + cw < C) ? cb + cw : C - 1;
      const unsigned zoff = (unsigned)(cc * ldz * 8) + 16u * lane;
#pragma unroll
      for (int k = 0; k < (MS_ROWS * E + 127) / 128; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_z, (__attribute__((address_space(3))) void*)(big + (cw * L::ZPITCH + 128 * k) * 8),
                                                 16, zoff, 1024 * k, 0, AUX);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int d = r + MS_ROWS * e;
    const double z0 = zs[(2 * j) * L::ZPITCH + d], z1 = zs[(2 * j + 1) * L::ZPITCH + d];
    dvec2 pn;
    pn.x = (a[e].x + eps * b[e].x) + s * z0;
    pn.y = (a[e].y + eps * b[e].y) + s * z1;
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(pn), r_thp, woff, e * slotb, AUX);
  }
}

}  // namespace

extern "C" {

int bk_mala_step_supported(int64_t C, int64_t D, int64_t ld) {
  // 32-bit byte offsets inside every array, with headroom for the slot offsets: arrays < 2 GiB
  return (C > 0 && D > 0 && D <= 1024 && C % 2 == 0 && ld % 2 == 0 && ld >= C && D * ld < ((int64_t)1 << 28)) ? 1 : 0;
}

int bk_mala_step(const double* theta, double* theta_out, double* grad, double* theta_prop,
                 const double* grad_prop, int64_t ld, double* lp, const double* lp_prop, const double* log_u,
                 const double* zt_next, int64_t ldz, double eps, double sqrt2eps, uint8_t* accept_mask,
                 double* ret, uint32_t* accept_count, int64_t C, int64_t D, void* stream) {
  if (!theta || !theta_out || !grad || !theta_prop || !grad_prop || !lp || !lp_prop || !log_u || C < 0 || D < 0)
    return BK_E_ARG;
  if (C == 0 || D == 0) return BK_OK;
  if (!bk_mala_step_supported(C, D, ld)) return BK_E_ALIGN;
  if (!bk_aligned16(theta) || !bk_aligned16(theta_out) || !bk_aligned16(grad) || !bk_aligned16(theta_prop) ||
      !bk_aligned16(grad_prop))
    return BK_E_ALIGN;
  if (zt_next && (ldz < D || ldz % 2 != 0 || !bk_aligned16(zt_next))) return BK_E_ALIGN;
  hipStream_t s = bk_stream(stream);
  dim3 grid((unsigned)bk_cdiv(C, MS_CHAINS)), block(MS_THREADS);
  const bool nt = bk_streams_past_llc(6 * C * D);
#define BK_MS_LAUNCH(E)                                                                                     \
  do {                                                                                                      \
    if (nt)                                                                                                 \
      k_mala_step<E, true><<<grid, block, 0, s>>>(theta, theta_out, grad, theta_prop, grad_prop, ld, lp,   \
                                                  lp_prop, log_u, zt_next, ldz, eps, sqrt2eps, accept_mask, \
                                                  ret, accept_count, C, D);                                 \
    else                                                                                                    \
      k_mala_step<E, false><<<grid, block, 0, s>>>(theta, theta_out, grad, theta_prop, grad_prop, ld, lp,  \
                                                   lp_prop, log_u, zt_next, ldz, eps, sqrt2eps,             \
                                                   accept_mask, ret, accept_count, C, D);                   \
  } while (0)
  if (D <= 128) BK_MS_LAUNCH(2);
  else if (D <= 256) BK_MS_LAUNCH(4);
  else if (D <= 512) BK_MS_LAUNCH(8);
  else BK_MS_LAUNCH(16);
#undef BK_MS_LAUNCH
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
