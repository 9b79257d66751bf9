// RNG-consuming kernels: one chain per lane, each lane walks its own NumPy-compatible
// stream sequentially over d (the reference consumes `rng.normal(size=D)` in index order,
// bayes_kit/hmc.py:56, mala.py:44, drghmc.py:360-364).  Stores are coalesced across the
// 64 lanes of a wavefront because the state layout is chain-contiguous.
#include "bk_common.hpp"
#include "bk_rng.hpp"
#include "ziggurat_tables.inc"

namespace {

__device__ const uint64_t d_zig_ki[256] = BK_ZIG_KI_INIT;
__device__ const uint64_t d_zig_wi[256] = BK_ZIG_WI_BITS_INIT;
__device__ const uint64_t d_zig_fi[256] = BK_ZIG_FI_BITS_INIT;
const uint64_t h_zig_ki[256] = BK_ZIG_KI_INIT;
const uint64_t h_zig_wi[256] = BK_ZIG_WI_BITS_INIT;
const uint64_t h_zig_fi[256] = BK_ZIG_FI_BITS_INIT;

// 6 KiB of ziggurat tables staged in LDS: the table index is lane-divergent (8 random
// bits), so LDS (64 banks) serves it far better than the scalar/constant path.
struct ZigLds {
  uint64_t ki[256];
  double wi[256];
  double fi[256];
};

__device__ __forceinline__ void load_tables(ZigLds& t) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    t.ki[i] = d_zig_ki[i];
    t.wi[i] = bk::u64_as_double(d_zig_wi[i]);
    t.fi[i] = bk::u64_as_double(d_zig_fi[i]);
  }
  __syncthreads();
}

constexpr int RNG_BLOCK = 64;  // one wavefront per workgroup: C/64 workgroups spread over the CUs
// When C is too small to give every SIMD a wavefront (256 CUs x 4 SIMDs x 64 lanes = 65,536
// lanes), half-filled wavefronts (32 chains per workgroup) reach twice as many SIMDs: these
// kernels are latency bound, so that is faster (measured 103 -> 90 us at 32,768 chains).
static inline int rng_block(i64 C) { return C <= 32768 ? 32 : 64; }

__global__ __launch_bounds__(256) void k_init_philox(uint64_t* st, i64 ldr, uint64_t key0,
                                                     uint64_t chain0, i64 C) {
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  st[0 * ldr + c] = key0;
  st[1 * ldr + c] = chain0 + (uint64_t)c;
  for (int w = 2; w < 10; ++w) st[w * ldr + c] = 0;
  st[10 * ldr + c] = 4;
}

template <typename G>
__global__ __launch_bounds__(RNG_BLOCK) void k_refresh(uint64_t* st, i64 ldr, const double* loc_in,
                                                       double loc_mul, double scale, double* out,
                                                       i64 ld, const double* metric, double* kin_out,
                                                       const uint8_t* active, i64 C, i64 D) {
  __shared__ ZigLds tab;
  load_tables(tab);
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (active && !active[c]) return;
  G g;
  g.load(st, ldr, c);
  // kinetic energy summed in the same order as bk_leapfrog_finish (four contiguous quarters of
  // the dimensions, each sequential, combined ((p0+p1)+p2)+p3), so the same momentum has the
  // same energy whichever kernel computes it
  const i64 Dq = (D + 3) / 4;
  double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;
  auto emit = [&](i64 d, double z) {
    double loc = loc_in ? loc_in[d * ld + c] * loc_mul : 0.0;
    double v = loc + scale * z;  // numpy random_normal: loc + scale * z
    out[d * ld + c] = v;
    if (kin_out) {
      double mv = metric ? metric[d] * v : v;
      double t = v * mv;
      if (d < Dq) k0 = k0 + t;
      else if (d < 2 * Dq) k1 = k1 + t;
      else if (d < 3 * Dq) k2 = k2 + t;
      else k3 = k3 + t;
    }
  };
  i64 d = 0;
  for (; d + 1 < D; d += 2) {  // two normals per pass (same stream order, more ILP)
    bk::top_up(g);
    double z0, z1;
    bk::next_normal_pair(g, tab.ki, tab.wi, tab.fi, z0, z1);
    emit(d, z0);
    emit(d + 1, z1);
  }
  if (d < D) {
    bk::top_up(g);
    emit(d, bk::next_normal(g, tab.ki, tab.wi, tab.fi));
  }
  g.store(st, ldr, c);
  if (kin_out) kin_out[c] = 0.5 * (((k0 + k1) + k2) + k3);
}

template <typename G, bool LOG>
__global__ __launch_bounds__(RNG_BLOCK) void k_log_uniform(uint64_t* st, i64 ldr, double* out,
                                                           const uint8_t* active, i64 C) {
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (active && !active[c]) return;
  G g;
  g.load(st, ldr, c);
  double u = bk::next_double(g);
  out[c] = LOG ? log(u) : u;
  g.store(st, ldr, c);
}

template <typename G>
__global__ __launch_bounds__(RNG_BLOCK) void k_mala_propose(uint64_t* st, i64 ldr, const double* theta,
                                                            const double* grad, double* prop, i64 ld,
                                                            double eps, double s, i64 C, i64 D) {
  __shared__ ZigLds tab;
  load_tables(tab);
  i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  G g;
  g.load(st, ldr, c);
  i64 d = 0;
  for (; d + 1 < D; d += 2) {
    bk::top_up(g);
    double z0, z1;
    bk::next_normal_pair(g, tab.ki, tab.wi, tab.fi, z0, z1);
    i64 o = d * ld + c;
    prop[o] = (theta[o] + eps * grad[o]) + s * z0;  // mala.py:41-45, left to right
    prop[o + ld] = (theta[o + ld] + eps * grad[o + ld]) + s * z1;
  }
  if (d < D) {
    bk::top_up(g);
    i64 o = d * ld + c;
    prop[o] = (theta[o] + eps * grad[o]) + s * bk::next_normal(g, tab.ki, tab.wi, tab.fi);
  }
  g.store(st, ldr, c);
}

// theta' = (theta + eps*grad) + s*z with z already drawn (mala.py:41-45), two rows per thread
__global__ __launch_bounds__(256) void k_mala_propose_z(const double* th, const double* g, const double* z,
                                                        double* prop, i64 ld, double eps, double s, i64 C, i64 D) {
  i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 d0 = (i64)blockIdx.y * 4;
  if (c >= C) return;
  double a[4], b[4], n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) {
      i64 o = (d0 + i) * ld + c;
      a[i] = th[o];
      b[i] = g[o];
      n[i] = z[o];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (d0 + i < D) prop[(d0 + i) * ld + c] = (a[i] + eps * b[i]) + s * n[i];
}

}  // namespace

extern "C" {

int bk_version(void) { return 100; }

int bk_rng_init_philox(uint64_t* state, int64_t ldr, uint64_t key0, uint64_t chain_id0, int64_t C,
                       void* stream) {
  if (!state || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  k_init_philox<<<dim3((unsigned)bk_cdiv(C, 256)), dim3(256), 0, bk_stream(stream)>>>(
      state, ldr, key0, chain_id0, C);
  BK_RETURN_LAUNCH_STATUS();
}

int bk_momentum_refresh(int rng_kind, uint64_t* state, int64_t ldr, const double* loc_in,
                        double loc_mul, double scale, double* out, int64_t ld, const double* metric,
                        double* kin_out, const uint8_t* active, int64_t C, int64_t D, void* stream) {
  if (!state || !out || C < 0 || D < 0 || ld < C || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_refresh<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, loc_in, loc_mul, scale, out,
                                                                ld, metric, kin_out, active, C, D);
  else if (rng_kind == BK_RNG_PCG64)
    k_refresh<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, loc_in, loc_mul, scale, out,
                                                               ld, metric, kin_out, active, C, D);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_log_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out, const uint8_t* active,
                   int64_t C, void* stream) {
  if (!state || !out || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_log_uniform<bk::Philox, true><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_log_uniform<bk::Pcg64, true><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out, const uint8_t* active, int64_t C,
               void* stream) {
  if (!state || !out || C < 0 || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_log_uniform<bk::Philox, false><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else if (rng_kind == BK_RNG_PCG64)
    k_log_uniform<bk::Pcg64, false><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, out, active, C);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_mala_propose(int rng_kind, uint64_t* state, int64_t ldr, const double* theta, const double* grad,
                    double* theta_prop, int64_t ld, double eps, double sqrt2eps, int64_t C, int64_t D,
                    void* stream) {
  if (!state || !theta || !grad || !theta_prop || C < 0 || D < 0 || ld < C || ldr < C) return BK_E_ARG;
  if (C == 0) return BK_OK;
  const int rb_ = rng_block(C);
  dim3 grid((unsigned)bk_cdiv(C, rb_)), block(rb_);
  if (rng_kind == BK_RNG_PHILOX)
    k_mala_propose<bk::Philox><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, theta, grad, theta_prop,
                                                                     ld, eps, sqrt2eps, C, D);
  else if (rng_kind == BK_RNG_PCG64)
    k_mala_propose<bk::Pcg64><<<grid, block, 0, bk_stream(stream)>>>(state, ldr, theta, grad, theta_prop,
                                                                    ld, eps, sqrt2eps, C, D);
  else
    return BK_E_ARG;
  BK_RETURN_LAUNCH_STATUS();
}

int bk_mala_propose_from_normals(const double* theta, const double* grad, const double* z, double* theta_prop,
                                 int64_t ld, double eps, double sqrt2eps, int64_t C, int64_t D, void* stream) {
  if (!theta || !grad || !z || !theta_prop || C < 0 || D < 0) return BK_E_ARG;
  if (ld < C) return BK_E_ALIGN;
  if (C == 0 || D == 0) return BK_OK;
  dim3 grid((unsigned)bk_cdiv(C, 256), (unsigned)bk_cdiv(D, 4));
  k_mala_propose_z<<<grid, dim3(256), 0, bk_stream(stream)>>>(theta, grad, z, theta_prop, ld, eps, sqrt2eps, C, D);
  BK_RETURN_LAUNCH_STATUS();
}

// ---- host self-test hooks: same source, host compilation ----------------------------
static void host_tables(const double** wi, const double** fi) {
  static double s_wi[256], s_fi[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; ++i) {
      s_wi[i] = bk::u64_as_double(h_zig_wi[i]);
      s_fi[i] = bk::u64_as_double(h_zig_fi[i]);
    }
    init = true;
  }
  *wi = s_wi;
  *fi = s_fi;
}

int bk_host_normals(int rng_kind, uint64_t* w, double* out, int64_t n) {
  if (!w || !out || n < 0) return BK_E_ARG;
  const double *wi, *fi;
  host_tables(&wi, &fi);
  if (rng_kind == BK_RNG_PHILOX) {
    bk::Philox g;
    g.load(w, (i64)1, (i64)0);
    i64 i = 0;
    for (; i + 1 < n; i += 2) bk::next_normal_pair(g, h_zig_ki, wi, fi, out[i], out[i + 1]);
    if (i < n) out[i] = bk::next_normal(g, h_zig_ki, wi, fi);
    g.store(w, (i64)1, (i64)0);
  } else if (rng_kind == BK_RNG_PCG64) {
    bk::Pcg64 g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_normal(g, h_zig_ki, wi, fi);
    g.store(w, (i64)1, (i64)0);
  } else {
    return BK_E_ARG;
  }
  return BK_OK;
}

int bk_host_uniforms(int rng_kind, uint64_t* w, double* out, int64_t n) {
  if (!w || !out || n < 0) return BK_E_ARG;
  if (rng_kind == BK_RNG_PHILOX) {
    bk::Philox g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_double(g);
    g.store(w, (i64)1, (i64)0);
  } else if (rng_kind == BK_RNG_PCG64) {
    bk::Pcg64 g;
    g.load(w, (i64)1, (i64)0);
    for (i64 i = 0; i < n; ++i) out[i] = bk::next_double(g);
    g.store(w, (i64)1, (i64)0);
  } else {
    return BK_E_ARG;
  }
  return BK_OK;
}

double bk_host_log1p(double x) { return bk::bk_log1p(x); }

}  // extern "C"
