// Convergence diagnostics on the device: streaming per-chain moments for R-hat
// (bayes_kit/rhat.py:111-171) and per-chain effective sample size (ess.py:52-69,
// iat.py:7-43,95-135, autocorr.py:6-33).
#include "bk_common.hpp"
#include "bk_welford.hpp"
#include <stdlib.h>

namespace {

using bkw::EL_ROWS;
using bkw::welford_elem;
constexpr int PC_BLOCK = 64;

__global__ __launch_bounds__(256) void k_welford(double* mean, double* m2, const double* th, i64 ld, i64 ld_th,
                                                 double n, const int64_t* n_dev, i64 n_off, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * EL_ROWS;
  if (c >= C) return;
  if (n_dev) n = (double)(*n_dev - n_off);
#pragma unroll
  for (int i = 0; i < EL_ROWS; ++i)
    if (d0 + i < D) {
      i64 o = (d0 + i) * ld + c;
      double mu = mean[o], q = m2[o];
      welford_elem(th[(d0 + i) * ld_th + c], mu, q, n);
      mean[o] = mu;
      m2[o] = q;
    }
}

// two chains (16 B) per lane (bkw::welford_unit_v2); n_dev != NULL: the update count is read from device memory -- n =
// *n_dev - n_off -- so that the launch can sit inside a sampler's captured draw
template <bool NT>
__global__ __launch_bounds__(256) void k_welford_v2(double* mean, double* m2, const double* th, i64 ld, i64 ld_th,
                                                    double n, const int64_t* n_dev, i64 n_off, i64 C2, i64 D) {
  if (n_dev) n = (double)(*n_dev - n_off);
  bkw::welford_unit_v2<NT>(blockIdx.x, blockIdx.y, (int)threadIdx.x, mean, m2, th, ld, ld_th, n, C2, D);
}

// one workgroup per dimension; thread t owns chains t, t+256, ... (fixed order), then a
// fixed-shape LDS tree: deterministic for a given C.
__global__ __launch_bounds__(256) void k_rhat_partials(const double* mean, const double* m2, i64 ld,
                                                       double nm1, const double* center, double* out,
                                                       i64 C, i64 D) {
  __shared__ double red[3][256];
  i64 d = blockIdx.x;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  double ctr = center ? center[d] : 0.0;
  for (i64 c = threadIdx.x; c < C; c += 256) {
    double mu = mean[d * ld + c];
    s0 = s0 + mu;
    s1 = s1 + m2[d * ld + c] / nm1;
    double dv = mu - ctr;
    s2 = s2 + dv * dv;
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
      red[2][threadIdx.x] += red[2][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0 * D + d] = red[0][0];
    out[1 * D + d] = red[1][0];
    if (center) out[2 * D + d] = red[2][0];
  }
}

// two-pass mean / variance per chain of a stored series (np.mean, np.var(ddof=1)): a workgroup of four wavefronts
// serves 64 chains (lane = chain); wavefront w sums the draws t = w, w + 4, w + 8, ... with eight loads in flight,
// the four partial sums are combined as ((p0 + p1) + p2) + p3.  (One lane walking its chain's N rows alone, one load
// at a time, ran at 0.5 TB/s: 1.09 ms for 1000 draws x 65,536 chains.)  len: per-chain draw counts (ragged chains).
__global__ __launch_bounds__(256) void k_chain_mean_var(const double* x, i64 ld, const int32_t* len, i64 N,
                                                        double* mean, double* var, i64 C) {
  __shared__ double part[4][BK_WAVE];
  const int lane = threadIdx.x & (BK_WAVE - 1), w = bk_wave_id();
  const i64 c = (i64)blockIdx.x * BK_WAVE + lane;
  const i64 n = c < C ? (len ? (i64)len[c] : N) : 0;
  const double* xc = x + (c < C ? c : 0);
  double s = 0.0;
  for (i64 t0 = w; t0 < N; t0 += 32) {  // (wavefront-uniform trip count; a lane's own draws end at n)
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (t0 + 4 * k < n) ? xc[(t0 + 4 * k) * ld] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = s + v[k];
  }
  part[w][lane] = s;
  __syncthreads();
  const double mu = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (double)n;
  __syncthreads();
  double q = 0.0;
  for (i64 t0 = w; t0 < N; t0 += 32) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (t0 + 4 * k < n) ? xc[(t0 + 4 * k) * ld] - mu : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) q = q + v[k] * v[k];
  }
  part[w][lane] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    mean[c] = mu;
    if (var) var[c] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (double)(n - 1);
  }
}

// ESS per chain, one lane per chain.  acor[n] = (sum_t xc[t]*xc[t+n]) / var0 / N with
// xc = x - mean, var0 = np.var(x) (ddof=0): the quantity autocorr.py:23-33 obtains via FFT.
// Lags are produced two at a time (the even/odd pair the Geyer estimators consume) and the
// scan stops at the first pair with negative sum (iat.py:38-43), so only the lags that
// matter are ever computed.
__global__ __launch_bounds__(PC_BLOCK) void k_ess(const double* x, i64 ld, i64 N, int estimator,
                                                  double* ess_out, double* iat_out, i64 C) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  const double* xc = x + c;
  double s = 0.0;
  for (i64 t = 0; t < N; ++t) s = s + xc[t * ld];
  double mu = s / (double)N;
  double q = 0.0;
  for (i64 t = 0; t < N; ++t) {
    double dv = xc[t * ld] - mu;
    q = q + dv * dv;
  }
  double var0 = q / (double)N;
  double total = 0.0, prev_min = 0.0;
  i64 n = 0;
  bool first = true;
  while (n + 1 < N) {
    double a0 = 0.0, a1 = 0.0;
    i64 t = 0;
    for (; t + n + 1 < N; ++t) {
      double u = xc[t * ld] - mu;
      a0 = a0 + u * (xc[(t + n) * ld] - mu);
      a1 = a1 + u * (xc[(t + n + 1) * ld] - mu);
    }
    a0 = a0 + (xc[t * ld] - mu) * (xc[(t + n) * ld] - mu);  // lag n has one more term
    double r0 = a0 / var0 / (double)N, r1 = a1 / var0 / (double)N;
    double pair = r0 + r1;
    if (first) {
      // iat.py:127-128: the first pair always enters the IMSE sum, even when negative
      prev_min = pair;
      if (estimator == 0) total = pair;
      first = false;
      if (pair < 0.0) break;  // n stays 0
      if (estimator == 1) total = pair;
    } else {
      if (pair < 0.0) break;
      if (estimator == 0) {
        prev_min = prev_min < pair ? prev_min : pair;  // iat.py:132
        total = total + prev_min;
      } else {
        total = total + pair;
      }
    }
    n += 2;
  }
  double iat = 2.0 * total - 1.0;
  if (iat_out) iat_out[c] = iat;
  ess_out[c] = (double)N / iat;
}

// all-lag autocorrelation (autocorr.py:6-33), one lane per chain, direct summation
__global__ __launch_bounds__(PC_BLOCK) void k_autocorr(const double* x, i64 ld, i64 N, double* out, i64 ldo,
                                                       i64 C) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  const double* xc = x + c;
  double s = 0.0;
  for (i64 t = 0; t < N; ++t) s = s + xc[t * ld];
  double mu = s / (double)N;
  double q = 0.0;
  for (i64 t = 0; t < N; ++t) {
    double dv = xc[t * ld] - mu;
    q = q + dv * dv;
  }
  double var0 = q / (double)N;
  for (i64 n = 0; n < N; ++n) {
    double a = 0.0;
    for (i64 t = 0; t + n < N; ++t) a = a + (xc[t * ld] - mu) * (xc[(t + n) * ld] - mu);
    out[n * ldo + c] = a / var0 / (double)N;
  }
}

// ---- ESS / autocorrelation with the series staged in LDS ----------------------------------------
// A workgroup owns G consecutive chains: their N draws are read once, in rows of G*8 contiguous
// bytes, into LDS (chain-major, pitch odd: the transposing writes are conflict-free), centred there,
// and every chain is then worked on by one WAVEFRONT: lane l owns lag n0 + l of a block of 64 lags
// and accumulates  a[n] = sum_t xc[t] * xc[t+n]  sequentially in t -- xc[t] is an LDS broadcast,
// xc[t+n0+l] a conflict-free consecutive read -- so 64 lags cost N steps, against 2 N strided
// global loads PER LAG in the one-lane-per-chain kernel above.  The Geyer scan (iat.py:38-43,
// :127-135) runs after each block and stops the chain at the first negative pair; typical chains
// need one block.  Per-lag summation order = the one-lane kernel's (sequential in t).
// Work per chain: O(N * ceil(lags/64)); all lags (bk_autocorr): O(N^2 / 64) per wavefront.
constexpr int ET_BLOCK = 256, ET_WAVES = ET_BLOCK / BK_WAVE;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < BK_WAVE; m <<= 1) v = v + __shfl_xor(v, m);
  return v;
}

// RT (series of ET_RT_MIN_DRAWS draws and more): a block of 64 lags is summed from REGISTER TILES instead of one LDS read
// per multiply-add.  Lane l owns 9 consecutive draws t = tc + 9 l + b of every chunk of 576 draws and keeps 64
// accumulators, one per lag of the block; a chunk needs the lane's 9 values x[t..t+8] and the window
// x[t + n0 .. t + n0 + 71] -- 81 LDS reads (stride 9 doubles between lanes: conflict-free) for 576 fused
// multiply-adds, against 1,152 reads.  The series is followed by >= 72 zeros in LDS, so draws and lags beyond the end
// need no masks.  A butterfly of 63 exchanges then leaves the total of lag n0 + l in lane l (the layout the Geyer scan
// below works on).  Per-lag order of summation: a lane's draws ascending, then a fixed tree over the lanes.
constexpr int ET_RT_MIN_DRAWS = 288, ET_RT_TAIL = 72;

template <int HALF>
__device__ __forceinline__ void ess_fold(double (&v)[64], int lane) {
  // lanes whose bit HALF is set keep the upper half of the first 2*HALF entries, the others the lower half
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const bool up = (lane & HALF) != 0;
    const double send = up ? v[i] : v[i + HALF];
    const double mine = up ? v[i + HALF] : v[i];
    v[i] = mine + __shfl_xor(send, HALF);
  }
}

template <int G, bool RT>
__global__ __launch_bounds__(ET_BLOCK) void k_ess_tile(const double* x, i64 ld, i64 N, int pitch, int estimator,
                                                       double* ess_out, double* iat_out, double* acor_out, i64 ldo,
                                                       i64 C) {
  extern __shared__ __attribute__((aligned(16))) double xs[];  // [G][pitch]
  const int t = threadIdx.x, lane = t & (BK_WAVE - 1), w = bk_wave_id();
  const i64 c0 = (i64)blockIdx.x * G;
  {
    const int cl = t % G, r0 = t / G;
    constexpr int RS = ET_BLOCK / G;
    const bool ok = c0 + cl < C;
    for (i64 r = r0; r < N; r += RS) xs[cl * pitch + r] = ok ? x[r * ld + c0 + cl] : 0.0;
    if (RT)
      for (i64 r = N + r0; r < pitch; r += RS) xs[cl * pitch + r] = 0.0;  // (the zeros behind the series)
  }
  __syncthreads();
  for (int cl = w; cl < G; cl += ET_WAVES) {
    const i64 c = c0 + cl;
    if (c >= C) break;  // wave-uniform
    double* xc = xs + cl * pitch;
    double s = 0.0;
    for (i64 r = lane; r < N; r += BK_WAVE) s = s + xc[r];
    const double mu = wave_sum(s) / (double)N;
    double q = 0.0;
    for (i64 r = lane; r < N; r += BK_WAVE) {
      const double dv = xc[r] - mu;
      xc[r] = dv;
      q = q + dv * dv;
    }
    const double var0 = wave_sum(q) / (double)N;
    // (each lane re-reads values other lanes centred: same wavefront, LDS ops are issued in order)
    double total = 0.0, prev_min = 0.0;
    bool first = true, done = false;
    for (i64 n0 = 0; n0 < N && !done; n0 += BK_WAVE) {
      const i64 n = n0 + lane;
      double a = 0.0;
      const i64 tmax = N - n0;  // lane 0's term count; lane l stops l terms earlier
      if (RT) {
        double acc[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) acc[j] = 0.0;
        for (i64 tc = 0; tc < tmax; tc += 9 * BK_WAVE) {
          const i64 t0 = tc + 9 * lane, ty = t0 + n0;
          const double* px = xc + (t0 < N ? t0 : N);  // (from N on: zeros)
          const double* py = xc + (ty < N ? ty : N);
          double X[9], win[16];
#pragma unroll
          for (int b = 0; b < 9; ++b) X[b] = px[b];
#pragma unroll
          for (int m = 0; m < 12; ++m) win[m] = py[m];
#pragma unroll
          for (int j = 0; j < 64; ++j) {
            if (j + 12 < ET_RT_TAIL) win[(j + 12) & 15] = py[j + 12];
#pragma unroll
            for (int b = 0; b < 9; ++b) acc[j] = __builtin_fma(X[b], win[(j + b) & 15], acc[j]);
          }
        }
        ess_fold<32>(acc, lane);
        ess_fold<16>(acc, lane);
        ess_fold<8>(acc, lane);
        ess_fold<4>(acc, lane);
        ess_fold<2>(acc, lane);
        ess_fold<1>(acc, lane);
        a = acc[0];
      }
      // main part: every lane's index is in range, 8 steps per pass with all 16 LDS reads issued before
      // the (ordered) accumulation -- one read-use dependency per step would cost an LDS latency each
      const i64 tsafe = tmax - (BK_WAVE - 1);  // tt + n0 + 63 < N  for tt < tsafe
      i64 tt = RT ? tmax : 0;
      for (; tt + 8 <= tsafe; tt += 8) {
        double u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          u[k] = xc[tt + k];
          v[k] = xc[tt + k + n];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a = a + u[k] * v[k];
      }
      for (; tt < tmax; ++tt) {
        const double u = xc[tt];
        const i64 j = tt + n;
        const double v = j < N ? xc[j] : 0.0;
        if (j < N) a = a + u * v;
      }
      const double r = n < N ? a / var0 / (double)N : 0.0;
      if (acor_out) {
        if (n < N) acor_out[n * ldo + c] = r;
        continue;  // all lags wanted: no truncation
      }
      const double pair = r + __shfl_down(r, 1);  // valid in even lanes
      for (int k = 0; k < BK_WAVE / 2; ++k) {
        if (n0 + 2 * k + 1 >= N) { done = true; break; }  // iat.py:38 `while n + 1 < N`
        const double pk = __shfl(pair, 2 * k);
        if (first) {
          // iat.py:127-128: the first pair always enters the IMSE sum, even when negative
          prev_min = pk;
          if (estimator == 0) total = pk;
          first = false;
          if (pk < 0.0) { done = true; break; }
          if (estimator == 1) total = pk;
        } else {
          if (pk < 0.0) { done = true; break; }
          if (estimator == 0) {
            prev_min = prev_min < pk ? prev_min : pk;  // iat.py:132
            total = total + prev_min;
          } else {
            total = total + pk;
          }
        }
      }
    }
    if (!acor_out && lane == 0) {
      const double iat = 2.0 * total - 1.0;
      if (iat_out) iat_out[c] = iat;
      ess_out[c] = (double)N / iat;
    }
  }
}

// IAT / ESS from an autocorrelation array acor[n*ld + c] (e.g. produced by an FFT for very long
// chains): the Geyer scan alone, one lane per chain, reads coalesced across chains.
__global__ __launch_bounds__(256) void k_iat_from_acor(const double* acor, i64 ld, i64 N, int estimator,
                                                       double* ess_out, double* iat_out, i64 C) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double total = 0.0, prev_min = 0.0;
  bool first = true;
  for (i64 n = 0; n + 1 < N; n += 2) {
    const double pk = acor[n * ld + c] + acor[(n + 1) * ld + c];
    if (first) {
      prev_min = pk;
      if (estimator == 0) total = pk;
      first = false;
      if (pk < 0.0) break;
      if (estimator == 1) total = pk;
    } else {
      if (pk < 0.0) break;
      if (estimator == 0) {
        prev_min = prev_min < pk ? prev_min : pk;
        total = total + prev_min;
      } else {
        total = total + pk;
      }
    }
  }
  const double iat = 2.0 * total - 1.0;
  if (iat_out) iat_out[c] = iat;
  ess_out[c] = (double)N / iat;
}

// index one past the last even-aligned pair (0,1), (2,3), ... before the first pair with a negative
// sum (iat.py:7-43), one lane per chain of an [N][ld] autocorrelation array
__global__ __launch_bounds__(PC_BLOCK) void k_end_pos_pairs(const double* acor, i64 ld, i64 N, i64* out, i64 C) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  i64 n = 0;
  while (n + 1 < N) {
    if (acor[n * ld + c] + acor[(n + 1) * ld + c] < 0.0) break;
    n += 2;
  }
  out[c] = n;
}

// Inverse standard-normal CDF: port of Cephes ndtri.c (Cephes Math Library, Copyright 1984-2000 by
// Stephen L. Moshier; redistributed by SciPy under BSD-3-Clause; see THIRD_PARTY.md) -- the routine
// scipy.stats.norm.ppf evaluates (rhat.py:106),
// same rational approximations and evaluation order, no FMA contraction.
__device__ __forceinline__ double polevl(double x, const double* c, int n) {
  double a = c[0];
  for (int i = 1; i <= n; ++i) a = a * x + c[i];
  return a;
}
__device__ __forceinline__ double p1evl(double x, const double* c, int n) {
  double a = x + c[0];
  for (int i = 1; i < n; ++i) a = a * x + c[i];
  return a;
}
__device__ double bk_ndtri(double y0) {
  const double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1,
                        1.39312609387279679503E1, -1.23916583867381258016E0};
  const double Q0[8] = {1.95448858338141759834E0, 4.67627912898881538453E0,  8.63602421390890590575E1,
                        -2.25462687854119370527E2, 2.00260212380060660359E2, -8.20372256168333339912E1,
                        1.59056225126211695515E1, -1.18331621121330003142E0};
  const double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1, 5.71628192246421288162E1,
                        4.40805073893200834700E1, 1.46849561928858024014E1, 2.18663306850790267539E0,
                        -1.40256079171354495875E-1, -3.50424626827848203418E-2, -8.57456785154685413611E-4};
  const double Q1[8] = {1.57799883256466749731E1, 4.53907635128879210584E1, 4.13172038254672030440E1,
                        1.50425385692907503408E1, 2.50464946208309415979E0, -1.42182922854787788574E-1,
                        -3.80806407691578277194E-2, -9.33259480895457427372E-4};
  const double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0,
                        1.33303460815807542389E0, 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                        3.01581553508235416007E-4, 2.65806974686737550832E-6, 6.23974539184983293730E-9};
  const double Q2[8] = {6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0,
                        2.16236993594496635890E-1, 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                        2.89247864745380683936E-6, 6.79019408009981274425E-9};
  const double expm2 = 0.13533528323661269189, s2pi = 2.50662827463100050242E0;
  if (y0 <= 0.0) return -INFINITY;
  if (y0 >= 1.0) return INFINITY;
  bool neg = true;
  double y = y0;
  if (y > 1.0 - expm2) {
    y = 1.0 - y;
    neg = false;
  }
  if (y > expm2) {
    y = y - 0.5;
    double y2 = y * y;
    double x = y + y * (y2 * polevl(y2, P0, 4) / p1evl(y2, Q0, 8));
    return x * s2pi;
  }
  double x = sqrt(-2.0 * log(y));
  double x0 = x - log(x) / x;
  double z = 1.0 / x;
  double x1 = x < 8.0 ? z * polevl(z, P1, 8) / p1evl(z, Q1, 8) : z * polevl(z, P2, 8) / p1evl(z, Q2, 8);
  x = x0 - x1;
  return neg ? -x : x;
}

// out[i] = ndtri((rank[i] - 0.325) / (S - 0.25))   (rhat.py:104-107; 0.325 as implemented there)
__global__ __launch_bounds__(256) void k_rank_normalize(const double* rank, double S, double* out, i64 n) {
  i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = bk_ndtri((rank[i] - 0.325) / (S - 0.25));
}

// ---- tracked series of a draw recorder ---------------------------------------------------------------------
// series[k][row][c] = theta[dims[k]][c] for k < K, series[K][row][c] = logp[c] (if given): one launch per draw
// instead of one strided copy per tracked coordinate.
__global__ __launch_bounds__(256) void k_record_series(const double* theta, i64 ld, const int32_t* dims, int K,
                                                       const double* logp, double* series, i64 cap, i64 row,
                                                       const int64_t* row_dev, i64 row_off, i64 C) {
  const i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (c >= C) return;
  if (row_dev) row = *row_dev - row_off;  // (the row index kept by the sampler's device-side draw counter)
  if (row < 0 || row >= cap) return;
  const double v = k < K ? theta[(i64)dims[k] * ld + c] : logp[c];
  series[((i64)k * cap + row) * C + c] = v;
}

}  // namespace

extern "C" {

static int welford_launch(double* mean, double* m2, i64 ld, const double* theta, i64 ld_th, i64 n, const int64_t* n_dev,
                          i64 n_off, i64 C, i64 D, void* stream) {
  if (!mean || !m2 || !theta || (!n_dev && n < 1) || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C || ld_th < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  if (bkw::v2_applies(mean, m2, theta, ld, ld_th, C)) {
    dim3 grid((unsigned)bk_cdiv(C / 2, 256), (unsigned)bk_cdiv(D, EL_ROWS));
    if (bk_streams_past_llc(3 * C * D))
      k_welford_v2<true><<<grid, dim3(256), 0, bk_stream(stream)>>>(mean, m2, theta, ld, ld_th, (double)n, n_dev, n_off,
                                                                    C / 2, D);
    else
      k_welford_v2<false><<<grid, dim3(256), 0, bk_stream(stream)>>>(mean, m2, theta, ld, ld_th, (double)n, n_dev, n_off,
                                                                     C / 2, D);
  } else {
    dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, EL_ROWS));
    k_welford<<<grid, dim3(256), 0, bk_stream(stream)>>>(mean, m2, theta, ld, ld_th, (double)n, n_dev, n_off, C, D);
  }
  BK_RETURN_LAUNCH_STATUS();
}

int bk_welford_update(double* mean, double* m2, int64_t ld, const double* theta, int64_t ld_theta, int64_t n,
                         int64_t C, int64_t D, void* stream) {
  return welford_launch(mean, m2, ld, theta, ld_theta, n, nullptr, 0, C, D, stream);
}

int bk_welford_update_dev(double* mean, double* m2, int64_t ld, const double* theta, int64_t ld_theta,
                          const int64_t* n_dev, int64_t n_offset, int64_t C, int64_t D, void* stream) {
  if (!n_dev) return BK_E_ARG;
  return welford_launch(mean, m2, ld, theta, ld_theta, 0, n_dev, n_offset, C, D, stream);
}

int bk_rhat_partials(const double* mean, const double* m2, int64_t ld, int64_t n, const double* center,
                     double* out, int64_t C, int64_t D, void* stream) {
  if (!mean || !m2 || !out || n < 2 || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (D == 0) return BK_OK;
  k_rhat_partials<<<dim3((unsigned)D), dim3(256), 0, bk_stream(stream)>>>(mean, m2, ld, (double)(n - 1), center,
                                                                        out, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_chain_mean_var(const double* x, int64_t ld, const int32_t* len, int64_t N, double* mean,
                      double* var, int64_t C, void* stream) {
  if (!x || !mean || N < 0 || C < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_chain_mean_var<<<dim3((unsigned)bk_cdiv(C, BK_WAVE)), dim3(256), 0, bk_stream(stream)>>>(x, ld, len, N, mean, var, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_rank_normalize(const double* rank, double S, double* out, int64_t n, void* stream) {
  if (!rank || !out || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_rank_normalize<<<dim3((unsigned)bk_cdiv(n, 256)), dim3(256), 0, bk_stream(stream)>>>(rank, S, out, n);
  BK_RETURN_LAUNCH_STATUS();
}

// LDS-staged launch for N draws: a chain group whose tile fits (G*pitch doubles <= 160 KiB less a margin); 0 if even
// one chain does not fit (the one-lane-per-chain kernels then serve).
static int ess_tile_launch(const double* x, i64 ld, i64 N, int estimator, double* ess_out, double* iat_out,
                           double* acor_out, i64 ldo, i64 C, hipStream_t s) {
  static const bool lane_only = []() { const char* e = getenv("BK_ESS_LANE_PER_CHAIN"); return e && e[0] == '1'; }();
  if (lane_only) return 0;  // experiments: the one-lane-per-chain kernels
  const bool rt = N >= ET_RT_MIN_DRAWS;
  const int pitch = (int)((rt ? N + ET_RT_TAIL : N) | 1);
  const i64 cap = (i64)(160 * 1024 - 512) / 8;
  // Two workgroups per CU when a tile allows it (one stages its chains while the other computes: 1000 draws x 65,536
  // chains 1.17 ms with 16 chains per workgroup, 0.68 with 8; 4000 draws 0.94 with 4, 0.74 with 2), else the widest
  // group that fits at all.
  const i64 cap2 = (i64)(78 * 1024) / 8;
  int G = 0;
  for (int g : {16, 8, 4, 2, 1})
    if ((i64)g * pitch <= cap2) { G = g; break; }
  if (!G)
    for (int g : {16, 8, 4, 2, 1})
      if ((i64)g * pitch <= cap) { G = g; break; }
  if (!G || N > 0x3fffffff) return 0;
  const size_t bytes = (size_t)G * pitch * 8;
  dim3 grid((unsigned)bk_cdiv(C, G)), block(ET_BLOCK);
#define BK_ET(GG)                                                                                                   \
  do {                                                                                                              \
    if (bytes > 64 * 1024) {                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rt ? &k_ess_tile<GG, true>                   \
                                                                          : &k_ess_tile<GG, false>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                   \
      if (e != hipSuccess) return -(int)e - 1000;                                                                   \
    }                                                                                                               \
    if (rt)                                                                                                         \
      k_ess_tile<GG, true><<<grid, block, bytes, s>>>(x, ld, N, pitch, estimator, ess_out, iat_out, acor_out, ldo,  \
                                                      C);                                                           \
    else                                                                                                            \
      k_ess_tile<GG, false><<<grid, block, bytes, s>>>(x, ld, N, pitch, estimator, ess_out, iat_out, acor_out, ldo, \
                                                       C);                                                          \
  } while (0)
  switch (G) {
    case 16: BK_ET(16); break;
    case 8: BK_ET(8); break;
    case 4: BK_ET(4); break;
    case 2: BK_ET(2); break;
    default: BK_ET(1); break;
  }
#undef BK_ET
  return 1;
}

int bk_autocorr(const double* x, int64_t ld, int64_t N, double* out, int64_t ldo, int64_t C, void* stream) {
  if (!x || !out || N < 2 || C < 0) return BK_E_ARG;
  if (ld < C || ldo < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  int r = ess_tile_launch(x, ld, N, 0, nullptr, nullptr, out, ldo, C, bk_stream(stream));
  if (r < 0) return -(r + 1000);
  if (r == 0)
    k_autocorr<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(x, ld, N, out, ldo, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_iat_from_acor(const double* acor, int64_t ld, int64_t N, int estimator, double* ess_out, double* iat_out,
                     int64_t C, void* stream) {
  if (!acor || !ess_out || N < 4 || C < 0 || (estimator != 0 && estimator != 1)) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_iat_from_acor<<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, bk_stream(stream)>>>(acor, ld, N, estimator,
                                                                                       ess_out, iat_out, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_end_pos_pairs(const double* acor, int64_t ld, int64_t N, int64_t* out, int64_t C, void* stream) {
  if (!out || N < 0 || C < 0 || (!acor && N > 0)) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_end_pos_pairs<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(acor, ld, N, out, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_ess(const double* x, int64_t ld, int64_t N, int estimator, double* ess_out, double* iat_out,
           int64_t C, void* stream) {
  if (!x || !ess_out || N < 4 || C < 0 || (estimator != 0 && estimator != 1)) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  int r = ess_tile_launch(x, ld, N, estimator, ess_out, iat_out, nullptr, 0, C, bk_stream(stream));
  if (r < 0) return -(r + 1000);
  if (r == 0)
    k_ess<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(x, ld, N, estimator,
                                                                                       ess_out, iat_out, C);
  BK_RETURN_LAUNCH_STATUS();
}

static int record_launch(const double* theta, i64 ld, const int32_t* dims, i64 K, const double* logp, double* series,
                         i64 capacity, i64 row, const int64_t* row_dev, i64 row_off, i64 C, void* stream) {
  if (!theta || !series || (K > 0 && !dims) || K < 0 || capacity < 1 || C < 0) return BK_E_ARG;
  if (!row_dev && (row < 0 || row >= capacity)) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  const int rows = (int)K + (logp ? 1 : 0);
  if (C == 0 || rows == 0) return BK_OK;
  k_record_series<<<dim3((unsigned)bk_cdiv(C, 256), (unsigned)rows), dim3(256), 0, bk_stream(stream)>>>(
      theta, ld, dims, (int)K, logp, series, capacity, row, row_dev, row_off, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_record_series(const double* theta, int64_t ld, const int32_t* dims, int64_t K, const double* logp,
                     double* series, int64_t capacity, int64_t row, int64_t C, void* stream) {
  return record_launch(theta, ld, dims, K, logp, series, capacity, row, nullptr, 0, C, stream);
}

int bk_record_series_dev(const double* theta, int64_t ld, const int32_t* dims, int64_t K, const double* logp,
                         double* series, int64_t capacity, const int64_t* row_dev, int64_t row_offset, int64_t C,
                         void* stream) {
  if (!row_dev) return BK_E_ARG;
  return record_launch(theta, ld, dims, K, logp, series, capacity, 0, row_dev, row_offset, C, stream);
}

}  // extern "C"
