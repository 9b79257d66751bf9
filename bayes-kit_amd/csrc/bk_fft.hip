// Autocorrelation of LONG stored series by FFT (bayes_kit/autocorr.py:23-33: zero-pad to
// S = 2**ceil(log2(2N-1)), |fft|^2, inverse, / var / N), for chains too long for the LDS-staged direct sums of
// bk_diag.hip.  Hand-written for the [N, C] layout of the stored series (draw-major, chains contiguous):
//
//   * The batch dimension is the contiguous one, so the transform runs ACROSS rows with one lane per column:
//     every lane of a wavefront executes the same butterfly on its own column, all loads and stores are
//     1-KiB row segments, the twiddles are wavefront-uniform (one table look-up through the scalar unit), no
//     LDS exchange, no bank conflicts.  Stockham autosort passes of radix 16 (8 / 4 / 2 for the remainder:
//     65,536 points are four passes) ping-pong between two scratch arrays: 32 bytes per element and pass --
//     HBM-bound, as the whole diagnostic is.
//   * Two real series per complex column: (x[t, 2c], x[t, 2c+1]) IS a complex number in this layout.  With
//     z = a + i b:  A[k] = (Z[k] + conj Z[S-k]) / 2,  B[k] = (Z[k] - conj Z[S-k]) / 2i, so the two power
//     spectra come from one transform, and because both autocovariances are real one inverse transform of
//     P_a + i P_b returns them as real and imaginary part: 2 complex transforms per PAIR of chains.
//     Each series is centred and scaled to unit variance on the way in (so a chain with a large variance does
//     not leak rounding error into its partner) -- the division by var of autocorr.py:32 done first.
//   * inverse = conj(forward(conj(.))) / S with the same kernels.
#include "bk_common.hpp"

namespace {

typedef double dvec2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ dvec2 cmul(dvec2 a, dvec2 b) {
  return (dvec2){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ void bfly2(dvec2& a, dvec2& b) {
  dvec2 t = a;
  a = t + b;
  b = t - b;
}
__device__ __forceinline__ dvec2 mul_mi(dvec2 a) { return (dvec2){a.y, -a.x}; }  // a * (-i)

// in-register DFTs, outputs in bit-reversed positions (dft_out<R>(r) = position of output r)
__device__ __forceinline__ void dft4(dvec2& u0, dvec2& u1, dvec2& u2, dvec2& u3) {
  bfly2(u0, u2);
  bfly2(u1, u3);
  u3 = mul_mi(u3);
  bfly2(u0, u1);
  bfly2(u2, u3);
}
// 16 = 4 x 4: input n = n1 + 4 n2, output k = 4 k1 + k2.  y[n1][k2] = DFT4 over n2, times W16^(n1 k2), then DFT4
// over n1.  With dft4's bit-reversed slots, y[n1][k2] sits at u[n1 + 4 br(k2)] and X[4 k1 + k2] at u[br(k1) + 4 br(k2)].
__device__ __forceinline__ void dft16(dvec2 (&u)[16]) {
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) dft4(u[n1], u[n1 + 4], u[n1 + 8], u[n1 + 12]);
  // slot 4 holds k2 = 2, slot 8 holds k2 = 1, slot 12 holds k2 = 3; W16^m = exp(-2 pi i m / 16)
  u[1 + 8] = cmul(u[1 + 8], (dvec2){c1, -s1});    // n1 k2 = 1
  u[2 + 8] = cmul(u[2 + 8], (dvec2){h, -h});      // 2
  u[3 + 8] = cmul(u[3 + 8], (dvec2){s1, -c1});    // 3
  u[1 + 4] = cmul(u[1 + 4], (dvec2){h, -h});      // 2
  u[2 + 4] = mul_mi(u[2 + 4]);                    // 4
  u[3 + 4] = cmul(u[3 + 4], (dvec2){-h, -h});     // 6
  u[1 + 12] = cmul(u[1 + 12], (dvec2){s1, -c1});  // 3
  u[2 + 12] = cmul(u[2 + 12], (dvec2){-h, -h});   // 6
  u[3 + 12] = cmul(u[3 + 12], (dvec2){-c1, s1});  // 9
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]);
}

template <int R>
__device__ __forceinline__ void dft(dvec2 (&u)[R]) {
  if constexpr (R == 2) {
    bfly2(u[0], u[1]);
  } else if constexpr (R == 4) {
    dft4(u[0], u[1], u[2], u[3]);
  } else if constexpr (R == 16) {
    dft16(u);
  } else {
    const double s = 0.70710678118654752440;
    bfly2(u[0], u[4]);
    bfly2(u[1], u[5]);
    bfly2(u[2], u[6]);
    bfly2(u[3], u[7]);
    u[5] = cmul(u[5], (dvec2){s, -s});  // W8^1
    u[6] = mul_mi(u[6]);               // W8^2
    u[7] = cmul(u[7], (dvec2){-s, -s}); // W8^3
    dft4(u[0], u[1], u[2], u[3]);
    dft4(u[4], u[5], u[6], u[7]);
  }
}
template <int R>
__device__ __forceinline__ int dft_out(int r) {
  if (R == 2) return r;
  if (R == 4) return ((r & 1) << 1) | (r >> 1);
  if (R == 8) return ((r & 1) << 2) | (r & 2) | (r >> 2);
  const int k1 = r >> 2, k2 = r & 3;  // 16
  return (((k1 & 1) << 1) | (k1 >> 1)) + 4 * (((k2 & 1) << 1) | (k2 >> 1));
}

// mean and 1/sd (ddof = 0) of each column in two sweeps (sum, then sum of squared deviations), each a partial sum
// per block of rows (lane = column, grid = column groups x row blocks) combined in a fixed order
constexpr int FFT_MOMENT_BLOCKS = 256;

template <bool SQ>
__global__ __launch_bounds__(64) void k_col_partial(const double* x, i64 ld, i64 N, i64 C, const double* mean,
                                                    double* part) {
  const i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const i64 rows = (N + gridDim.y - 1) / gridDim.y, lo = (i64)blockIdx.y * rows, hi = (lo + rows < N) ? lo + rows : N;
  const double mu = SQ ? mean[c] : 0.0;
  double s = 0.0;
  for (i64 t = lo; t < hi; ++t) {
    const double d = x[t * ld + c] - mu;
    s += SQ ? d * d : d;
  }
  part[(i64)blockIdx.y * C + c] = s;
}

template <bool SQ>
__global__ __launch_bounds__(64) void k_col_combine(const double* part, int blocks, i64 N, i64 C, double* out) {
  const i64 c = (i64)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += part[(i64)b * C + c];
  out[c] = SQ ? 1.0 / sqrt(s / (double)N) : s / (double)N;
}

// W[m] = exp(-2 pi i m / S)
__global__ __launch_bounds__(256) void k_twiddles(dvec2* W, i64 S) {
  i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
  if (m >= S) return;
  double sn, cs;
  sincospi(2.0 * (double)m / (double)S, &sn, &cs);
  W[m] = (dvec2){cs, -sn};
}

// One Stockham pass of radix R over S rows x Cp complex columns: butterfly j (of T = S / R) reads rows
// j + r T, multiplies by W^(r k S / (Ns R)) with k = j mod Ns, and writes the R outputs to rows
// (j - k) R + k + r Ns.  One wavefront per butterfly and 64 columns.
//   MODE 0: complex in -> complex out
//   MODE 1: first forward pass (Ns = 1): rows come from the real series, two per column, centred and scaled,
//           rows >= N are the zero padding
//   MODE 2: first pass of the inverse (Ns = 1): rows are conj(P_a + i P_b), the two power spectra formed from
//           rows k and S - k of the forward transform
template <int R, int MODE>
__global__ __launch_bounds__(256) void k_fft_pass(const dvec2* in, dvec2* out, const dvec2* W, i64 S, i64 Ns, i64 Cp, i64 ldc,
                                                  const double* x, i64 ldx, i64 N, i64 C, const double* mean,
                                                  const double* inv_sd) {
  const int lane = threadIdx.x & 63;
  const i64 T = S / R;
  const i64 j = __builtin_amdgcn_readfirstlane((int)((i64)blockIdx.x * 4 + bk_wave_id()));
  const i64 cp = (i64)blockIdx.y * 64 + lane;
  if (j >= T || cp >= Cp) return;
  const i64 k = j & (Ns - 1);
  dvec2 u[R];
  if (MODE == 1) {
    const i64 ca = 2 * cp, cb = 2 * cp + 1;
    // (a column without a finite 1/sd -- constant, or holding a nan -- goes in as zeros: its result is nan
    // (k_fft_finish), and it must not reach its partner through the complex arithmetic)
    const double ma = mean[ca], sa = isfinite(inv_sd[ca]) ? inv_sd[ca] : 0.0;
    const double mb = cb < C ? mean[cb] : 0.0, sb = (cb < C && isfinite(inv_sd[cb])) ? inv_sd[cb] : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i64 t = j + r * T;
      u[r] = (dvec2){0.0, 0.0};
      if (t < N) {
        if (sa != 0.0) u[r].x = (x[t * ldx + ca] - ma) * sa;
        if (sb != 0.0) u[r].y = (x[t * ldx + cb] - mb) * sb;
      }
    }
  } else if (MODE == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i64 t = j + r * T;
      const dvec2 z = in[t * ldc + cp], zr = in[((S - t) & (S - 1)) * ldc + cp];
      const dvec2 p = {z.x + zr.x, z.y - zr.y}, m = {z.x - zr.x, z.y + zr.y};  // Z + conj Zr, Z - conj Zr
      u[r] = (dvec2){0.25 * (p.x * p.x + p.y * p.y), -(0.25 * (m.x * m.x + m.y * m.y))};
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = in[(j + r * T) * ldc + cp];
    const i64 step = k * (S / (Ns * R));  // (wavefront-uniform: the table look-ups go through the scalar unit)
#pragma unroll
    for (int r = 1; r < R; ++r) u[r] = cmul(u[r], W[r * step]);
  }
  dft<R>(u);
  const i64 j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) out[(j0 + r * Ns) * ldc + cp] = u[dft_out<R>(r)];
}

// rows 0..N-1 of the inverse transform: out[t, 2c] = Re / (S N), out[t, 2c+1] = -Im / (S N); 0 / 0 for a column
// of zero variance, as autocorr.py:32 has it
__global__ __launch_bounds__(256) void k_fft_finish(const dvec2* buf, i64 Cp, i64 ldc, double* out, i64 ldo, i64 N,
                                                    i64 C, double scale, const double* inv_sd) {
  const i64 cp = (i64)blockIdx.y * 256 + threadIdx.x, t = blockIdx.x;
  if (cp >= Cp) return;
  const dvec2 v = buf[t * ldc + cp];
  const double nan = __builtin_nan("");
  out[t * ldo + 2 * cp] = isfinite(inv_sd[2 * cp]) ? v.x * scale : nan;
  if (2 * cp + 1 < C) out[t * ldo + 2 * cp + 1] = isfinite(inv_sd[2 * cp + 1]) ? -v.y * scale : nan;
}

// Scratch: two [S][ldc] complex arrays, the twiddle table, mean / 1/sd, the row-block partial sums.  The row pitch is
// kept 576 bytes off a multiple of 4 KiB: a butterfly's R rows are S/R rows apart, and on a power-of-two pitch they
// (and the rows of the neighbouring butterflies) would all sit on the same memory channels.
struct FftPlan {
  i64 S, Cp, ldc;
  int blocks;
  size_t off_b, off_w, off_mean, off_isd, off_part, bytes;
};

FftPlan fft_plan(i64 N, i64 C) {
  FftPlan p;
  p.S = 1;
  while (p.S < 2 * N - 1) p.S <<= 1;
  p.Cp = (C + 1) / 2;
  const i64 mod = p.Cp >= 1024 ? 256 : 64;  // (narrow batches: 576 B off a 1-KiB multiple, less scratch wasted)
  // (fewer than 64 pairs of chains -- the reference's own call shape is ONE chain -- get no padding: a row is at most
  // 1 KiB, there is nothing to spread, and a pad to 36 columns made one chain of 4M draws a 10 GB plan)
  p.ldc = p.Cp < 64 ? p.Cp : p.Cp + ((36 - p.Cp % mod) + mod) % mod;
  p.blocks = (int)(N / 64 < 1 ? 1 : (N / 64 > FFT_MOMENT_BLOCKS ? FFT_MOMENT_BLOCKS : N / 64));
  const size_t buf = (size_t)p.S * (size_t)p.ldc * sizeof(dvec2);
  p.off_b = buf;
  p.off_w = 2 * buf;
  p.off_mean = p.off_w + (size_t)p.S * sizeof(dvec2);
  p.off_isd = p.off_mean + (size_t)(2 * p.Cp) * sizeof(double);
  p.off_part = p.off_isd + (size_t)(2 * p.Cp) * sizeof(double);
  p.bytes = p.off_part + (size_t)p.blocks * (size_t)(2 * p.Cp) * sizeof(double);
  return p;
}

template <int MODE>
void launch_pass(int R, const dvec2* in, dvec2* out, const dvec2* W, const FftPlan& p, i64 Ns, const double* x, i64 ldx,
                 i64 N, i64 C, const double* mean, const double* isd, hipStream_t s) {
  const dim3 block(256), grid((unsigned)bk_cdiv(p.S / R, 4), (unsigned)bk_cdiv(p.Cp, 64));
  if (R == 16)
    k_fft_pass<16, MODE><<<grid, block, 0, s>>>(in, out, W, p.S, Ns, p.Cp, p.ldc, x, ldx, N, C, mean, isd);
  else if (R == 8)
    k_fft_pass<8, MODE><<<grid, block, 0, s>>>(in, out, W, p.S, Ns, p.Cp, p.ldc, x, ldx, N, C, mean, isd);
  else if (R == 4)
    k_fft_pass<4, MODE><<<grid, block, 0, s>>>(in, out, W, p.S, Ns, p.Cp, p.ldc, x, ldx, N, C, mean, isd);
  else
    k_fft_pass<2, MODE><<<grid, block, 0, s>>>(in, out, W, p.S, Ns, p.Cp, p.ldc, x, ldx, N, C, mean, isd);
}

}  // namespace

extern "C" {

int64_t bk_autocorr_fft_work_bytes(int64_t N, int64_t C) {
  if (N < 2 || C < 1) return 0;
  return (int64_t)fft_plan(N, C).bytes;
}

int bk_autocorr_fft(const double* x, int64_t ld, int64_t N, double* out, int64_t ldo, int64_t C, void* work,
                    int64_t work_bytes, void* stream) {
  if (!x || !out || N < 2 || C < 0 || ld < C || ldo < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const FftPlan p = fft_plan(N, C);
  if (!work || work_bytes < (int64_t)p.bytes || !bk_aligned16(work)) return BK_E_ARG;
  if (p.S > 0x40000000 || bk_cdiv(p.Cp, 64) > 65535) return BK_E_ARG;
  hipStream_t s = bk_stream(stream);
  char* base = static_cast<char*>(work);
  dvec2* bufs[2] = {reinterpret_cast<dvec2*>(base), reinterpret_cast<dvec2*>(base + p.off_b)};
  dvec2* W = reinterpret_cast<dvec2*>(base + p.off_w);
  double* mean = reinterpret_cast<double*>(base + p.off_mean);
  double* isd = reinterpret_cast<double*>(base + p.off_isd);
  double* part = reinterpret_cast<double*>(base + p.off_part);
  {
    const dim3 gp((unsigned)bk_cdiv(C, 64), (unsigned)p.blocks), gc((unsigned)bk_cdiv(C, 64));
    k_col_partial<false><<<gp, dim3(64), 0, s>>>(x, ld, N, C, nullptr, part);
    k_col_combine<false><<<gc, dim3(64), 0, s>>>(part, p.blocks, N, C, mean);
    k_col_partial<true><<<gp, dim3(64), 0, s>>>(x, ld, N, C, mean, part);
    k_col_combine<true><<<gc, dim3(64), 0, s>>>(part, p.blocks, N, C, isd);
  }
  k_twiddles<<<dim3((unsigned)bk_cdiv(p.S, 256)), dim3(256), 0, s>>>(W, p.S);
  // radices of the passes: 16 while that leaves nothing or at least a factor 2 for a last pass of 8 / 4 / 2
  int radix[64], n_pass = 0;
  for (i64 rem = p.S; rem > 1;) {
    int R = rem >= 16 ? 16 : (int)rem;
    radix[n_pass++] = R;
    rem /= R;
  }
  int cur = 0;  // index of the buffer holding the current data
  for (int dir = 0; dir < 2; ++dir) {
    i64 Ns = 1;
    for (int q = 0; q < n_pass; ++q) {
      const int R = radix[q];
      if (q == 0 && dir == 0) {
        launch_pass<1>(R, nullptr, bufs[0], W, p, Ns, x, ld, N, C, mean, isd, s);
        cur = 0;
      } else if (q == 0) {
        launch_pass<2>(R, bufs[cur], bufs[cur ^ 1], W, p, Ns, x, ld, N, C, mean, isd, s);
        cur ^= 1;
      } else {
        launch_pass<0>(R, bufs[cur], bufs[cur ^ 1], W, p, Ns, x, ld, N, C, mean, isd, s);
        cur ^= 1;
      }
      Ns *= R;
    }
  }
  k_fft_finish<<<dim3((unsigned)N, (unsigned)bk_cdiv(p.Cp, 256)), dim3(256), 0, s>>>(
      bufs[cur], p.Cp, p.ldc, out, ldo, N, C, 1.0 / ((double)p.S * (double)N), isd);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
