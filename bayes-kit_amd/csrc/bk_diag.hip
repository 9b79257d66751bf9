// Convergence diagnostics on the device: streaming per-chain moments for R-hat
// (bayes_kit/rhat.py:111-171) and per-chain effective sample size (ess.py:52-69,
// iat.py:7-43,95-135, autocorr.py:6-33).
#include "bk_common.hpp"

namespace {

constexpr int EL_ROWS = 4;
constexpr int PC_BLOCK = 64;

// Welford: after the n-th draw, mean = np.mean(draws[:n]) and m2/(n-1) = np.var(ddof=1)
__global__ __launch_bounds__(256) void k_welford(double* mean, double* m2, const double* th, i64 ld,
                                                 double n, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * EL_ROWS;
  if (c >= C) return;
#pragma unroll
  for (int i = 0; i < EL_ROWS; ++i)
    if (d0 + i < D) {
      i64 o = (d0 + i) * ld + c;
      double x = th[o], mu = mean[o];
      double delta = x - mu;
      mu = mu + delta / n;
      mean[o] = mu;
      m2[o] = m2[o] + delta * (x - mu);
    }
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

// two-pass mean / variance per chain of a stored series (np.mean, np.var(ddof=1))
__global__ __launch_bounds__(PC_BLOCK) void k_chain_mean_var(const double* x, i64 ld, const int32_t* len,
                                                             i64 N, double* mean, double* var, i64 C) {
  i64 c = (i64)blockIdx.x * PC_BLOCK + threadIdx.x;
  if (c >= C) return;
  i64 n = len ? (i64)len[c] : N;
  double s = 0.0;
  for (i64 t = 0; t < n; ++t) s = s + x[t * ld + c];
  double mu = s / (double)n;
  double q = 0.0;
  for (i64 t = 0; t < n; ++t) {
    double dv = x[t * ld + c] - mu;
    q = q + dv * dv;
  }
  mean[c] = mu;
  if (var) var[c] = q / (double)(n - 1);
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

// Inverse standard-normal CDF: Cephes ndtri (what scipy.stats.norm.ppf evaluates; rhat.py:106),
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

}  // namespace

extern "C" {

int bk_welford_update(double* mean, double* m2, const double* theta, int64_t ld, int64_t n, int64_t C,
                      int64_t D, void* stream) {
  if (!mean || !m2 || !theta || n < 1 || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, EL_ROWS));
  k_welford<<<grid, dim3(256), 0, bk_stream(stream)>>>(mean, m2, theta, ld, (double)n, C, D);
  BK_RETURN_LAUNCH_STATUS();
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
  k_chain_mean_var<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(
      x, ld, len, N, mean, var, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_rank_normalize(const double* rank, double S, double* out, int64_t n, void* stream) {
  if (!rank || !out || n < 0) return BK_E_ARG;
  if (n == 0) return BK_OK;
  k_rank_normalize<<<dim3((unsigned)bk_cdiv(n, 256)), dim3(256), 0, bk_stream(stream)>>>(rank, S, out, n);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_autocorr(const double* x, int64_t ld, int64_t N, double* out, int64_t ldo, int64_t C, void* stream) {
  if (!x || !out || N < 2 || C < 0) return BK_E_ARG;
  if (ld < C || ldo < C) return BK_E_ALIGN;
  if (C == 0) return BK_OK;
  k_autocorr<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(x, ld, N, out, ldo, C);
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
  k_ess<<<dim3((unsigned)bk_cdiv(C, PC_BLOCK)), dim3(PC_BLOCK), 0, bk_stream(stream)>>>(x, ld, N, estimator,
                                                                                     ess_out, iat_out, C);
  BK_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
