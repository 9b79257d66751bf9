// Dense mass matrix: Y = M @ X for every chain at once, on the fp64 matrix cores.
//
// The reference only has a diagonal "metric" (bayes_kit/hmc.py:22,37,46-52); a dense one has
// no reference counterpart (parity unpinned).  The driver (hmc.py,
// metric_dense=) uses three matrix applications per chain: rho = chol(M) @ z, the kick
// rho += eps * (M @ grad), and the kinetic energy 1/2 rho . (M^-1 @ rho).  With the
// chain-contiguous layout, applying a matrix to all chains is one GEMM
// Y[D x C] = M[D x D] * X[D x C] -- the only GEMM-shaped work on this path, so it is the
// only place that uses MFMA: v_mfma_f64_16x16x4_f64 (2048 flop, 64 cycles per SIMD).
//
// Tiling: 256 threads = 4 wavefronts in 2 x 2, workgroup tile 128 rows x 128 chains, each
// wavefront 64 x 64 = 4 x 4 MFMA tiles (16 independent accumulators: the matrix pipe never
// waits on a dependent accumulate).  K advances 16 at a time through double-buffered LDS
// panels; the fp64 MFMA is slow enough (1024 cycles of matrix work per wavefront per K=4)
// that LDS and global traffic hide completely.  LDS rows are padded to 144 doubles so the
// four 16-lane k-groups of a fragment read land on disjoint banks.  Workgroups that share
// a panel of X (same chains, different rows of M) are placed on the same XCD back to back
// so the panel is fetched from HBM once and then served by that XCD's L2.
#include "bk_common.hpp"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double dvec2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDP = 144;  // padded panel pitch (doubles): 1152 B = 128 B mod 256 B

struct alignas(16) DenseLds {
  double a[2][BK][LDP];  // [k][row of M]
  double b[2][BK][LDP];  // [k][chain]
};

// Thread -> panel elements (the same for the global loads and the LDS stores):
//   A (128 rows of M x 16 k, k contiguous in global): thread t holds row (t & 127), k = 8 (t >> 7) .. + 8.
//     Its eight LDS stores go to a[k][row]: 16 consecutive lanes = 16 consecutive rows = 32 distinct banks
//     (the k pitch of 288 dwords is a multiple of the 32 store banks, so lanes must not differ in k only:
//     round 2's (row = t >> 1, k = 8 (t & 1)) mapping made every store a 2-way conflict).
//   B (16 k x 128 chains, chain contiguous): thread t holds k = t >> 4 and, for i = 0..3, the chain pair
//     32 i + 2 (t & 15): a global load instruction reads 256 contiguous bytes per k, and the 16-byte LDS
//     store of 8 consecutive lanes covers 32 consecutive dwords (round 2 stored 8 consecutive chains per
//     thread with 8-byte stores: lanes 128 B apart, an 8-way conflict; SQ_LDS_BANK_CONFLICT was 57 % of the
//     LDS cycles of the kernel, profiles/r3_dense_mfma.md).
template <bool FULL>
__device__ __forceinline__ void load_panels(const double* M, i64 ldm, const double* X, i64 ld, i64 r0,
                                            i64 c0, i64 k0, i64 R, i64 D, i64 C, double (&ra)[8],
                                            double (&rb)[8]) {
  const int t = threadIdx.x;
  const int row = t & 127, ka = (t >> 7) * 8;
  const int kb = t >> 4, q = (t & 15) * 2;
  if (FULL) {  // whole tiles, 16-B aligned rows: four 16-B loads per panel, no bounds checks
    const dvec2* pa = reinterpret_cast<const dvec2*>(M + (r0 + row) * ldm + k0 + ka);
    const double* pb = X + (k0 + kb) * ld + c0 + q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dvec2 va = pa[i], vb = *reinterpret_cast<const dvec2*>(pb + 32 * i);
      ra[2 * i] = va.x;
      ra[2 * i + 1] = va.y;
      rb[2 * i] = vb.x;
      rb[2 * i + 1] = vb.y;
    }
    return;
  }
  {
    i64 r = r0 + row;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      i64 k = k0 + ka + i;
      ra[i] = (r < R && k < D) ? M[r * ldm + k] : 0.0;
    }
  }
  {
    i64 k = k0 + kb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      i64 c = c0 + 32 * (i >> 1) + q + (i & 1);
      rb[i] = (k < D && c < C) ? X[k * ld + c] : 0.0;
    }
  }
}

__device__ __forceinline__ void store_panels(DenseLds& lds, int buf, const double (&ra)[8], const double (&rb)[8]) {
  const int t = threadIdx.x;
  {
    const int row = t & 127, ka = (t >> 7) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) lds.a[buf][ka + i][row] = ra[i];
  }
  {
    const int kb = t >> 4, q = (t & 15) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dvec2 v = {rb[2 * i], rb[2 * i + 1]};
      *reinterpret_cast<dvec2*>(&lds.b[buf][kb][32 * i + q]) = v;
    }
  }
}

// EPI 1: the result leaves as  y_rows[r] - sigmoid(z)  (the logistic regression's residual, the arithmetic of
// k_logistic_residual): the elementwise pass between the target's two GEMMs rides on the first one's epilogue --
// its exp and division run on the vector pipe while the CU's other workgroup keeps the matrix pipe busy.
template <bool FULL, int EPI>
__global__ __launch_bounds__(256, 2) void k_dense_apply(const double* M, i64 ldm, const double* X, double* Y,
                                                        i64 ld, i64 ldy, i64 C, i64 R, i64 Dtot, int row_blocks,
                                                        int chain_blocks, i64 k_chunk, i64 slab_elems,
                                                        const double* y_rows) {
  __shared__ DenseLds lds;
  // split-K: blockIdx.y owns the inner-dimension range [k_lo, k_hi) and its own output slab
  const i64 k_lo = (i64)blockIdx.y * k_chunk;
  const i64 D = (k_lo + k_chunk < Dtot) ? k_lo + k_chunk : Dtot;  // exclusive upper bound of k
  Y += (i64)blockIdx.y * slab_elems;
  // XCD-aware placement: consecutive slots of one XCD walk the row blocks of one chain block
  const int nblk = row_blocks * chain_blocks;
  int id = blockIdx.x;
  int rb_i, cb_i;
  {
    const int xcds = 8;
    int per = (chain_blocks + xcds - 1) / xcds;  // chain blocks per XCD (last ones may be empty)
    int xcd = id % xcds, slot = id / xcds;
    cb_i = (slot / row_blocks) * xcds + xcd;
    rb_i = slot % row_blocks;
    (void)per;
    (void)nblk;
    if (cb_i >= chain_blocks) return;
  }
  const i64 r0 = (i64)rb_i * BM, c0 = (i64)cb_i * BN;
  const int lane = threadIdx.x & 63, w = bk_wave_id();
  const int wr = (w >> 1) * 64, wc = (w & 1) * 64;  // wavefront's 64 x 64 corner inside the tile
  const int l15 = lane & 15, l4 = lane >> 4;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  double ra[8], rb[8];
  load_panels<FULL>(M, ldm, X, ld, r0, c0, k_lo, R, D, C, ra, rb);
  store_panels(lds, 0, ra, rb);
  __syncthreads();
  const i64 nk = (D - k_lo + BK - 1) / BK;
  for (i64 kb = 0; kb < nk; ++kb) {
    const int buf = (int)(kb & 1);
    if (kb + 1 < nk) load_panels<FULL>(M, ldm, X, ld, r0, c0, k_lo + (kb + 1) * BK, R, D, C, ra, rb);  // in flight under the MFMAs
    // fragments of k-slice ks + 4 are read while the MFMAs of slice ks run (two register sets)
    double a[2][4], b[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = lds.a[buf][l4][wr + 16 * i + l15];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[0][j] = lds.b[buf][l4][wc + 16 * j + l15];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const int cur = (ks / 4) & 1;
      if (ks + 4 < BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[cur ^ 1][i] = lds.a[buf][ks + 4 + l4][wr + 16 * i + l15];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[cur ^ 1][j] = lds.b[buf][ks + 4 + l4][wc + 16 * j + l15];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      if (ks == BK / 2 - 4 && kb + 1 < nk) {
        // the next panels go to the other LDS buffer (last read one iteration ago, behind a barrier) while
        // the matrix pipe still has half of this panel's MFMAs queued: the stores and the wait for the global
        // loads overlap matrix work instead of following it
        store_panels(lds, buf ^ 1, ra, rb);
      }
    }
    if (kb + 1 < nk) __syncthreads();
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      i64 c = c0 + wc + 16 * j + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        i64 r = r0 + wr + 16 * i + l4 + 4 * v;
        if (FULL || (r < R && c < C)) {
          double out = acc[i][j][v];
          if (EPI == 1) {
            const double e = exp(-fabs(out));
            const double p = out >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
            out = y_rows[r] - p;
          }
          if (k_chunk >= 0) Y[r * ldy + c] = out;
        }
      }
    }
}

// out[c] = scale * sum_d x[d,c] * y[d,c], one lane per chain, sequential in d
__global__ __launch_bounds__(64) void k_dot_columns(const double* x, const double* y, i64 ld, double scale,
                                                    double* out, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  constexpr int U = 8;
  for (i64 d0 = 0; d0 < D; d0 += U) {
    double a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (d0 + u < D) {
        a[u] = x[(d0 + u) * ld + c];
        b[u] = y[(d0 + u) * ld + c];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (d0 + u < D) s = s + a[u] * b[u];
  }
  out[c] = scale * s;
}

// deterministic reduction of split-K slabs: Y = sum_s P[s] in increasing s (slab rows of pitch ldw, Y rows of pitch ldy)
__global__ __launch_bounds__(256) void k_sum_slabs(const double* P, i64 slab, int S, i64 ldw, double* Y, i64 ldy, i64 R,
                                                   i64 C) {
  const i64 c = (i64)blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= C || r >= R) return;
  const i64 i = r * ldw + c;
  double a = P[i];
  for (int s = 1; s < S; ++s) a = a + P[(i64)s * slab + i];
  Y[r * ldy + c] = a;
}

}  // namespace
namespace {

// one GEMM launch (+ one for a last partial block of rows): `out` rows of pitch ldy, split-K slab q at out + q * slab_elems
static int gemm_block(const double* A, i64 lda, i64 R, i64 K, const double* X, i64 ldx, double* out, i64 ldy, i64 C,
                      i64 S, i64 k_chunk, i64 slab_elems, void* stream, const double* y_rows) {
  i64 rb = bk_cdiv(R, BM), cb = bk_cdiv(C, BN);
  i64 per = (cb + 7) / 8;
  if (per * 8 * rb > 0x7fffffff) return BK_E_ARG;
  const bool full_but_rows = (K % BK == 0) && (C % BN == 0) && (lda % 2 == 0) && (ldx % 2 == 0) && bk_aligned16(A) &&
                             bk_aligned16(X);
  hipStream_t st = bk_stream(stream);
  // Whole 128-row blocks take the unchecked kernel; a last partial block of rows (R = 10^6 observations: 64 rows)
  // is a second, small launch of the checked one instead of putting bounds checks into every tile of the first.
  const i64 R_full = full_but_rows ? R - R % BM : 0;
  if (R_full > 0) {
    const i64 rbf = R_full / BM;
    const dim3 grid((unsigned)(per * 8 * rbf), (unsigned)S);
    if (y_rows)
      k_dense_apply<true, 1><<<grid, dim3(256), 0, st>>>(A, lda, X, out, ldx, ldy, C, R_full, K, (int)rbf, (int)cb,
                                                         k_chunk, slab_elems, y_rows);
    else
      k_dense_apply<true, 0><<<grid, dim3(256), 0, st>>>(A, lda, X, out, ldx, ldy, C, R_full, K, (int)rbf, (int)cb,
                                                         k_chunk, slab_elems, nullptr);
  }
  if (R_full < R) {
    // rows [R_full, R): its split-K slabs sit at the same row offset inside each slab of R rows
    const i64 rr = R - R_full, rbr = bk_cdiv(rr, BM);
    const dim3 grid((unsigned)(per * 8 * rbr), (unsigned)S);
    if (y_rows)
      k_dense_apply<false, 1><<<grid, dim3(256), 0, st>>>(A + R_full * lda, lda, X, out + R_full * ldy, ldx, ldy, C, rr, K,
                                                          (int)rbr, (int)cb, k_chunk, slab_elems, y_rows + R_full);
    else
      k_dense_apply<false, 0><<<grid, dim3(256), 0, st>>>(A + R_full * lda, lda, X, out + R_full * ldy, ldx, ldy, C, rr, K,
                                                          (int)rbr, (int)cb, k_chunk, slab_elems, nullptr);
  }
  BK_RETURN_LAUNCH_STATUS();
}

// Split of a long inner dimension: how many slabs, and how many k per slab.  A function of (R, K) ONLY -- never of
// the number of chains in the call -- so that a chain's result does not depend on how many chains share its launch
// (a shard of 256 chains must reproduce its slice of a 2,048-chain run bit for bit).  The factor is
// what fills 256 CUs x 2 workgroups at the reference width of GEMM_COLS chains; wider calls are cut into column
// blocks of that width (which also bounds the slab scratch), narrower ones run with fewer tiles.
constexpr i64 GEMM_COLS = 2048;
static i64 gemm_slabs(i64 R, i64 K, i64* k_chunk_out) {
  const i64 rb = bk_cdiv(R, BM), cb_ref = GEMM_COLS / BN;
  i64 S = 1;
  if (rb * cb_ref < 1024 && K >= 64 * BK) {
    S = bk_cdiv(2048, rb * cb_ref);
    const i64 maxS = K / (32 * BK);  // keep >= 32 K-panels per split
    if (S > maxS) S = maxS;
    if (S > 65535) S = 65535;
    if (S < 2) S = 1;
  }
  i64 k_chunk = K;
  if (S > 1) {
    k_chunk = bk_cdiv(bk_cdiv(K, S), BK) * BK;
    S = bk_cdiv(K, k_chunk);
  }
  if (k_chunk_out) *k_chunk_out = k_chunk;
  return S;
}

static int gemm_launch(const double* A, i64 lda, i64 R, i64 K, const double* X, i64 ldx, double* Y, i64 ldy, i64 C,
                       double* work, i64 work_elems, void* stream, const double* y_rows = nullptr) {
  if (!A || !X || !Y || C < 0 || R < 0 || K < 0 || lda < K) return BK_E_ARG;
  if (ldx < C || ldy < C) return BK_E_ALIGN;
  if (C == 0 || R == 0) return BK_OK;
  i64 k_chunk = K, S = 1;
  if (work && !y_rows) S = gemm_slabs(R, K, &k_chunk);
  if (S > 1) {
    // column blocks of GEMM_COLS chains, each through the same S slabs of pitch ldw (the scratch is reused: the
    // launches are stream-ordered)
    const i64 ldw = C < GEMM_COLS ? C + (C & 1) : GEMM_COLS;
    if (work_elems < S * R * ldw) return BK_E_ARG;  // (a smaller S would change the bits: refuse instead)
    for (i64 c0 = 0; c0 < C; c0 += GEMM_COLS) {
      const i64 cw = C - c0 < GEMM_COLS ? C - c0 : GEMM_COLS;
      int rc = gemm_block(A, lda, R, K, X + c0, ldx, work, ldw, cw, S, k_chunk, R * ldw, stream, nullptr);
      if (rc != BK_OK) return rc;
      k_sum_slabs<<<dim3((unsigned)bk_cdiv(cw, 256), (unsigned)R), dim3(256), 0, bk_stream(stream)>>>(
          work, R * ldw, (int)S, ldw, Y + c0, ldy, R, cw);
    }
    BK_RETURN_LAUNCH_STATUS();
  }
  return gemm_block(A, lda, R, K, X, ldx, Y, ldy, C, 1, K, 0, stream, y_rows);
}

}  // namespace

extern "C" {

int bk_dense_metric_apply(const double* M, int64_t ldm, const double* X, double* Y, int64_t ld, int64_t C,
                          int64_t D, void* stream) {
  return gemm_launch(M, ldm, D, D, X, ld, Y, ld, C, nullptr, 0, stream);
}

int64_t bk_gemm_chains_work_elems(int64_t R, int64_t K, int64_t C) {
  if (R < 1 || K < 1 || C < 1) return 0;
  const i64 S = gemm_slabs(R, K, nullptr);
  if (S < 2) return 0;
  const i64 ldw = C < GEMM_COLS ? C + (C & 1) : GEMM_COLS;
  return S * R * ldw;
}

int bk_gemm_chains(const double* A, int64_t lda, int64_t R, int64_t K, const double* X, int64_t ldx, double* Y,
                   int64_t ldy, int64_t C, double* work, int64_t work_elems, void* stream) {
  return gemm_launch(A, lda, R, K, X, ldx, Y, ldy, C, work, work_elems, stream);
}

int bk_gemm_chains_logistic(const double* A, int64_t lda, int64_t R, int64_t K, const double* X, int64_t ldx,
                            double* Y, int64_t ldy, int64_t C, const double* y_rows, void* stream) {
  if (!y_rows) return BK_E_ARG;
  return gemm_launch(A, lda, R, K, X, ldx, Y, ldy, C, nullptr, 0, stream, y_rows);
}

int bk_dot_columns(const double* x, const double* y, int64_t ld, double scale, double* out, int64_t C, int64_t D,
                   void* stream) {
  if (!x || !y || !out || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_dot_columns<<<dim3((unsigned)bk_cdiv(C, 64)), dim3(64), 0, bk_stream(stream)>>>(x, y, ld, scale, out, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"

// ---- logistic regression target (config 5): the elementwise part between the two GEMMs --------
namespace {

// Z[n][c] (= x_n . theta_c) -> residual y_n - sigmoid(z) in place; per-segment partial sums of
// the log likelihood  y z - log(1 + e^z)  (softplus evaluated stably), summed in row order.  LL false: the caller
// wants the gradient only (every leapfrog step but the last): no log1p, nothing written to part.
// The pass is 16 B per element at one exp + one division (+ one log1p) each: RL_ROWS rows are requested before the
// first is used (one load in flight per thread ran at 2.5 TB/s: 13 ms for the 32.8 GB of config 5).
constexpr int RL_ROWS = 8;
template <bool LL>
__global__ __launch_bounds__(256) void k_logistic_residual(double* Z, i64 ldz, const double* y, double* part,
                                                           i64 N, i64 C, i64 rows_per_seg) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 seg = blockIdx.y;
  if (c >= C) return;
  i64 n0 = seg * rows_per_seg, n1 = n0 + rows_per_seg;
  if (n1 > N) n1 = N;
  double ll = 0.0;
  for (i64 nb = n0; nb < n1; nb += RL_ROWS) {
    double zz[RL_ROWS];
#pragma unroll
    for (int i = 0; i < RL_ROWS; ++i)
      if (nb + i < n1) zz[i] = Z[(nb + i) * ldz + c];
#pragma unroll
    for (int i = 0; i < RL_ROWS; ++i) {
      if (nb + i < n1) {
        const double z = zz[i];
        const double yn = y[nb + i];
        const double e = exp(-fabs(z));
        if (LL) {
          const double sp = (z > 0.0 ? z : 0.0) + log1p(e);  // log(1 + e^z)
          ll = ll + (yn * z - sp);
        }
        const double p = z >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
        Z[(nb + i) * ldz + c] = yn - p;
      }
    }
  }
  if (LL) part[seg * C + c] = ll;
}

// grad = t * G + (-(inv_s2 * theta)); logp = t * sum_s part[s] + (-0.5 * inv_s2 * sum theta^2)
__global__ __launch_bounds__(64) void k_logistic_finish(const double* G, const double* th, i64 ld, const double* part,
                                                        i64 S, double inv_s2, double t, double* grad, double* logp,
                                                        double* ll_out, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s2 = 0.0;
  for (i64 d = 0; d < D; ++d) {
    double x = th[d * ld + c];
    s2 = s2 + x * x;
    if (grad) grad[d * ld + c] = t * G[d * ld + c] + (-(inv_s2 * x));
  }
  double ll = 0.0;
  if (part)
    for (i64 s = 0; s < S; ++s) ll = ll + part[s * C + c];
  if (ll_out) ll_out[c] = ll;
  if (logp) logp[c] = t * ll + (-0.5 * inv_s2 * s2);
}

}  // namespace

extern "C" {

int bk_logistic_residual(double* Z, int64_t ldz, const double* y, double* part, int64_t N, int64_t C,
                         int64_t segments, void* stream) {
  if (!Z || !y || N < 0 || C < 0 || segments < 1 || segments > 65535) return BK_E_ARG;
  if (ldz < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  i64 rows = bk_cdiv(N > 0 ? N : 1, segments);
  dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)segments);
  if (part) k_logistic_residual<true><<<grid, dim3(256), 0, bk_stream(stream)>>>(Z, ldz, y, part, N, C, rows);
  else k_logistic_residual<false><<<grid, dim3(256), 0, bk_stream(stream)>>>(Z, ldz, y, nullptr, N, C, rows);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_logistic_finish(const double* G, const double* theta, int64_t ld, const double* part, int64_t segments,
                       double inv_prior_var, double t, double* grad, double* logp, double* loglik, int64_t C,
                       int64_t D, void* stream) {
  if (!theta || (!part && (logp || loglik)) || (grad && !G) || C < 0 || D < 0 || segments < 1) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_logistic_finish<<<dim3((unsigned)bk_cdiv(C, 64)), dim3(64), 0, bk_stream(stream)>>>(
      G, theta, ld, part, segments, inv_prior_var, t, grad, logp, loglik, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
