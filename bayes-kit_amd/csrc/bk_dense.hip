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

struct DenseLds {
  double a[2][BK][LDP];  // [k][row of M]
  double b[2][BK][LDP];  // [k][chain]
};

template <bool FULL>
__device__ __forceinline__ void load_panels(const double* M, i64 ldm, const double* X, i64 ld, i64 r0,
                                            i64 c0, i64 k0, i64 D, i64 C, double (&ra)[8], double (&rb)[8]) {
  const int t = threadIdx.x;
  if (FULL) {  // whole tiles, 16-B aligned rows: four 16-B loads per panel, no bounds checks
    const dvec2* pa = reinterpret_cast<const dvec2*>(M + (r0 + (t >> 1)) * ldm + k0 + (t & 1) * 8);
    const dvec2* pb = reinterpret_cast<const dvec2*>(X + (k0 + (t >> 4)) * ld + c0 + (t & 15) * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dvec2 va = pa[i], vb = pb[i];
      ra[2 * i] = va.x;
      ra[2 * i + 1] = va.y;
      rb[2 * i] = vb.x;
      rb[2 * i + 1] = vb.y;
    }
    return;
  }
  // A panel: 128 rows x 16 k, row-major in global (k contiguous): thread -> (row, 8 consecutive k)
  {
    int row = t >> 1, kk = (t & 1) * 8;
    i64 r = r0 + row;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      i64 k = k0 + kk + i;
      ra[i] = (r < D && k < D) ? M[r * ldm + k] : 0.0;
    }
  }
  // B panel: 16 k x 128 chains (chain contiguous): thread -> (k, 8 consecutive chains)
  {
    int kk = t >> 4, cc = (t & 15) * 8;
    i64 k = k0 + kk;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      i64 c = c0 + cc + i;
      rb[i] = (k < D && c < C) ? X[k * ld + c] : 0.0;
    }
  }
}

__device__ __forceinline__ void store_panels(DenseLds& lds, int buf, const double (&ra)[8], const double (&rb)[8]) {
  const int t = threadIdx.x;
  {
    int row = t >> 1, kk = (t & 1) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) lds.a[buf][kk + i][row] = ra[i];
  }
  {
    int kk = t >> 4, cc = (t & 15) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) lds.b[buf][kk][cc + i] = rb[i];
  }
}

template <bool FULL>
__global__ __launch_bounds__(256, 2) void k_dense_apply(const double* M, i64 ldm, const double* X, double* Y,
                                                        i64 ld, i64 C, i64 D, int row_blocks, int chain_blocks) {
  __shared__ DenseLds lds;
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
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = (w >> 1) * 64, wc = (w & 1) * 64;  // wavefront's 64 x 64 corner inside the tile
  const int l15 = lane & 15, l4 = lane >> 4;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  double ra[8], rb[8];
  load_panels<FULL>(M, ldm, X, ld, r0, c0, 0, D, C, ra, rb);
  store_panels(lds, 0, ra, rb);
  __syncthreads();
  const i64 nk = (D + BK - 1) / BK;
  for (i64 kb = 0; kb < nk; ++kb) {
    const int buf = (int)(kb & 1);
    if (kb + 1 < nk) load_panels<FULL>(M, ldm, X, ld, r0, c0, (kb + 1) * BK, D, C, ra, rb);  // in flight under the MFMAs
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = lds.a[buf][ks + l4][wr + 16 * i + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = lds.b[buf][ks + l4][wc + 16 * j + l15];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nk) {
      store_panels(lds, buf ^ 1, ra, rb);
      __syncthreads();
    }
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
        if (FULL || (r < D && c < C)) Y[r * ld + c] = acc[i][j][v];
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

}  // namespace

extern "C" {

int bk_dense_metric_apply(const double* M, int64_t ldm, const double* X, double* Y, int64_t ld, int64_t C,
                          int64_t D, void* stream) {
  if (!M || !X || !Y || C < 0 || D < 0 || ldm < D) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  int row_blocks = (int)bk_cdiv(D, BM), chain_blocks = (int)bk_cdiv(C, BN);
  int per = (chain_blocks + 7) / 8;
  unsigned grid = (unsigned)(per * 8 * row_blocks);
  bool full = (D % BM == 0) && (C % BN == 0) && (ldm % 2 == 0) && (ld % 2 == 0) && bk_aligned16(M) && bk_aligned16(X);
  if (full)
    k_dense_apply<true><<<dim3(grid), dim3(256), 0, bk_stream(stream)>>>(M, ldm, X, Y, ld, C, D, row_blocks,
                                                                        chain_blocks);
  else
    k_dense_apply<false><<<dim3(grid), dim3(256), 0, bk_stream(stream)>>>(M, ldm, X, Y, ld, C, D, row_blocks,
                                                                         chain_blocks);
  BK_RETURN_LAUNCH_STATUS();
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
