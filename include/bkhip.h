/* bkhip.h -- C ABI of libbkhip.so: the MI355X (gfx950) many-chain HMC / MALA / DRGHMC
 * hot path that stands in for the per-draw arithmetic of flatironinstitute/bayes-kit.
 *
 * Conventions (all entry points)
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - Every pointer is a DEVICE pointer owned by the caller (the library never allocates
 *     persistent memory and never frees caller memory).
 *   - Phase-space arrays are fp64, chain-major-contiguous ("[D][ld]"): element (d, c) of
 *     a D-dimensional state of chain c lives at  p[d*ld + c],  ld >= C.  One chain per
 *     GPU lane, so a wavefront touches 64 consecutive doubles for every d.
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work and return.
 *   - Return value: 0 = ok; < 0 = argument error (BK_E_*); > 0 = hipError_t of the launch.
 *   - No global mutable state: re-entrant across host threads and one-process-per-GPU.
 *   - Arithmetic is IEEE fp64 with every * and + individually rounded (the library is
 *     built with -ffp-contract=off) in the operation order of the cited reference lines,
 *     so elementwise results are bit-identical to the reference's NumPy expressions.
 *   - Two FORMS of every streaming kernel, one entry point: an even number of chains, an even ld and 16-byte aligned
 *     pointers get the 16-bytes-per-lane form (two chains per lane: `k_*_v2` in the sources -- what the many-chain
 *     samplers always pass); an odd chain count, an odd pitch or an unaligned view gets the 8-byte form (`k_*`), same
 *     arithmetic, same results.  The split is by SHAPE, decided inside the entry point (bk_leapfrog_kick_drift,
 *     bk_leapfrog_finish, bk_select_columns, bk_blend_columns, bk_welford_update, the Gaussian targets); the `_n`
 *     variants are the 8-byte form sized by a device-side lane count.  Neither is a superseded generation of the other.
 *
 * Each entry point cites the reference code (flatironinstitute/bayes-kit) it replaces.
 */
#ifndef BKHIP_H
#define BKHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_OK 0
#define BK_E_ARG (-1)      /* null pointer / bad extent / bad enum */
#define BK_E_ALIGN (-2)    /* ld < C or similar layout violation */

/* ---- per-chain random streams ------------------------------------------------------
 * state: u64 table [BK_RNG_WORDS][ldr], one chain per column.
 *   BK_RNG_PHILOX (np.random.Philox(key=[k0,k1])): rows 0-1 key, 2-5 counter,
 *       6-9 output buffer, 10 buffer_pos (4 = empty) -- the fields of numpy's
 *       bit_generator.state, so a reference stream can be resumed mid-way.
 *   BK_RNG_PCG64 (np.random.default_rng(int)): rows 0-1 state (hi, lo), 2-3 inc (hi, lo).
 */
#define BK_RNG_WORDS 11
#define BK_RNG_PHILOX 0
#define BK_RNG_PCG64 1

int bk_version(void);

/* Philox key = (key0, chain_id0 + c), counter 0, buffer empty: the stream of
 * np.random.Generator(np.random.Philox(key=[key0, chain_id0 + c])).  Replaces
 * `self._rng = np.random.default_rng(seed)` (hmc.py:23, mala.py:25, drghmc.py:71) for
 * chain c when the reference is seeded with that Philox bit generator. */
int bk_rng_init_philox(uint64_t* state, int64_t ldr, uint64_t key0, uint64_t chain_id0,
                       int64_t C, void* stream);

/* out[d*ld + c] = loc + scale * z,  z ~ N(0,1) drawn d = 0..D-1 in order from chain c's
 * stream (NumPy ziggurat; `rng.normal(loc, scale, size=D)`), where
 *     loc = loc_in ? loc_in[d*ld + c] * loc_mul : 0.0
 * and, if kin_out != NULL,  kin_out[c] = 0.5 * sum_d out*(metric[d]*out)  (metric NULL = 1).
 *   HMC momentum draw + kinetic term     hmc.py:56, :37      (loc_in NULL, scale 1)
 *   initial theta / rho                  hmc.py:24-28, drghmc.py:72-77
 *   DRGHMC partial refresh               drghmc.py:360-364, :250  (loc_mul = sqrt(1-damping),
 *                                                                  scale = sqrt(damping))
 * `active` (NULL = all): u8 mask per chain; inactive chains draw nothing.
 * `work` (may be NULL): caller-owned scratch of bk_refresh_work_elems(C, D) doubles.  With it
 * (Philox streams, no mask, D >= 32) the normals are produced by one WAVEFRONT per chain -- 256
 * stream words evaluated at once, the sequential consumption order resolved exactly -- instead
 * of one lane per chain: same stream, same values and final state, C wavefronts of parallelism
 * instead of C/64 (measured: 4.9x faster at 4096 chains x 128 dims, 17x at 1024 x 1024,
 * 1.3x at 65,536 x 1024 including the transpose into [D][C]). */
int64_t bk_refresh_work_elems(int64_t C, int64_t D);
int bk_momentum_refresh(int rng_kind, uint64_t* state, int64_t ldr,
                        const double* loc_in, double loc_mul, double scale,
                        double* out, int64_t ld, const double* metric, double* kin_out,
                        const uint8_t* active, int64_t C, int64_t D, double* work,
                        int64_t work_elems, void* stream);

/* DRGHMC: the partial momentum refresh with its kinetic energy (bk_momentum_refresh, no mask) followed by the
 * start of the draw (bk_dr_begin_retry with kin = kin_out) -- drghmc.py:360-371.  With `work` (Philox, D >= 32) the
 * transpose of the normals into the state layout, the kinetic energy and the start of the draw are ONE launch after
 * the generator's; otherwise the two calls in sequence.  Same values, same stream positions either way.
 * side (may be NULL): the PREVIOUS draw's diagnostics -- one bk_welford_update_dev call (mean NULL: none) and / or one
 * bk_record_series_dev call (series NULL: none) on the chains' current point as this draw finds it -- done by workgroups of
 * the generator's launch: the generator is bound by instruction issue, the Welford update by memory, and side by side they
 * take the longer of the two times instead of the sum (a draw sequence inside one hipGraph: DrGhmcDiag.advance(n)).  Where
 * the generator's launch cannot carry them (no `work`, odd C, ...) they are launches of their own, first. */
typedef struct bk_diag_job {
  const double* theta; /* [D][ld_theta]: the point both parts read */
  int64_t ld_theta;
  int64_t C, D;
  const int64_t* n_dev; /* the sampler's device-side draw count */
  double* mean;         /* bk_welford_update_dev(mean, m2, ld, theta, ld_theta, n_dev, n_offset, C, D) */
  double* m2;
  int64_t ld;
  int64_t n_offset;
  double* series; /* bk_record_series_dev(theta, ld_theta, dims, K, logp, series, capacity, n_dev, row_offset, C) */
  const int32_t* dims;
  int64_t K;
  const double* logp;
  int64_t capacity;
  int64_t row_offset;
} bk_diag_job;
int bk_dr_refresh_begin(int rng_kind, uint64_t* state, int64_t ldr, const double* loc_in, double loc_mul,
                        double scale, double* out, int64_t ld, const double* metric, double* kin_out, int64_t C,
                        int64_t D, double* work, int64_t work_elems, const double* logp, double* cur_H,
                        double* cur_h, double* rej, uint8_t* alive, double prob_retry, uint32_t* counters,
                        int64_t n_counters, int64_t* draw_counter, const bk_diag_job* side, void* stream);

/* out[c] = log(u), u = next double of chain c's stream: `np.log(self._rng.uniform())`
 * (hmc.py:60, metropolis.py:74, drghmc.py:370,378).  Inactive chains draw nothing and
 * keep out[c]. */
int bk_log_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out,
                   const uint8_t* active, int64_t C, void* stream);

/* out[c] = u itself (`rng.uniform()`): the resampling uniforms of smc.py:73. */
int bk_uniform(int rng_kind, uint64_t* state, int64_t ldr, double* out, const uint8_t* active,
               int64_t C, void* stream);

/* ---- leapfrog integrator -------------------------------------------------------------
 * One fused kick + drift over all chains and dimensions (the hot loop hmc.py:47-49,
 * drghmc.py:277-278,282-283):
 *     t      = metric[d] * grad            (metric NULL = ones, still multiplied by 1.0)
 *     r      = rho_in
 *     if use_pre:  r = r + pre  * t        (hmc.py:46 back half-step, pre = -(0.5*eps);
 *                                           drghmc.py:277 first half-kick, pre = 0.5*h)
 *     if use_kick: r = r + kick * t        (hmc.py:48 / drghmc.py:282, kick = eps)
 *     rho_out   = r
 *     theta_out = theta_in + eps * r       (hmc.py:49 / drghmc.py:278,283)
 * theta_out may alias theta_in and rho_out may alias rho_in (in-place), or differ (the
 * first step of a trajectory reads the current point and writes the proposal buffers).
 * grad may have any strides (ldg_d, ldg_c): element (d,c) at grad[d*ldg_d + c*ldg_c]; the
 * chain-contiguous case (ldg_c == 1) streams directly, the dimension-contiguous case
 * (ldg_d == 1, what a row-major (C, D) model output looks like) is transposed through
 * LDS tiles.  Algorithmic HBM bytes: 40 per element. */
int bk_leapfrog_kick_drift(const double* theta_in, double* theta_out,
                           const double* rho_in, double* rho_out, int64_t ld,
                           const double* grad, int64_t ldg_d, int64_t ldg_c,
                           const double* metric, double eps,
                           int use_pre, double pre, int use_kick, double kick,
                           int64_t C, int64_t D, void* stream);

/* Same step, but chain j of the outputs (compact, leading dimension ld_out) is chain
 * src_index[j] of the inputs (leading dimension ld_in): gathers the active chains of a
 * delayed-rejection stage into a dense buffer while taking their first leapfrog step
 * (drghmc.py:276-278 for the chains that reach proposal k). n = number of gathered chains. */
int bk_leapfrog_first_step_gather(const double* theta_in, const double* rho_in, const double* grad_in,
                                  int64_t ld_in, const int32_t* src_index, double* theta_out,
                                  double* rho_out, int64_t ld_out, const double* metric, double eps,
                                  double pre, int64_t n, int64_t D, const uint32_t* n_dev, void* stream);
/* (n_dev may be NULL: all n lanes; otherwise min(n, *n_dev), see "Lane counts on the device" below.) */

/* Final half-kick and kinetic energy of a trajectory:
 *     r = rho_in + half * (metric[d] * grad)      hmc.py:52, drghmc.py:286
 *     if negate: r = -r                           drghmc.py:345 (momentum flip)
 *     if rho_out: rho_out = r
 *     kin_out[c] = 0.5 * sum_d r*(metric[d]*r)    hmc.py:37, drghmc.py:250
 * grad NULL: no kick (r = rho_in), i.e. just the kinetic energy of a finished trajectory.
 * Per-chain sums (here, in bk_momentum_refresh, bk_mala_logq and the Gaussian targets' logp)
 * run over four contiguous quarters of the dimensions, each sequentially in d, combined as
 * ((p0+p1)+p2)+p3: a fixed order that depends on D only (np.dot uses yet another order:
 * results agree to ~1e-16 relative, see DESIGN.md tolerances). */
int bk_leapfrog_finish(const double* rho_in, double* rho_out, int64_t ld,
                       const double* grad, int64_t ldg_d, int64_t ldg_c,
                       const double* metric, double half, int negate,
                       double* kin_out, int64_t C, int64_t D, void* stream);

/* ---- the same three steps over a lane set whose SIZE LIVES ON THE DEVICE -----------------------
 * (drghmc.py:276-278, :280-283, :285-286 for the chains that reach a delayed-rejection proposal or one
 * of its ghosts.)  n_dev: device pointer to the number of lanes in the set, written by an earlier launch
 * on the same stream (bk_compact_indices, the appending accept tests); the launch is sized for the bound
 * C / n and works on min(bound, *n_dev) lanes -- see "Lane counts on the device" below.  With these and a
 * counted gradient (bk_target_*_grad_n, bk_target_fn_n) a delayed-rejection draw over ANY model that the
 * library can call without the host is a fixed launch sequence: no lane count is read back, the draw
 * replays as one hipGraph.  n_dev NULL = the host-sized entry points above.  Same arithmetic. */
int bk_leapfrog_kick_drift_n(const double* theta_in, double* theta_out, const double* rho_in, double* rho_out,
                             int64_t ld, const double* grad, int64_t ldg_d, int64_t ldg_c,
                             const double* metric, double eps, int use_pre, double pre, int use_kick,
                             double kick, int64_t C, int64_t D, const uint32_t* n_dev, void* stream);
/* bk_leapfrog_finish for such a set, plus what the end of a delayed-rejection trajectory owes its level:
 * H_out / h_out / live_out (all or none; needs logp = the log density at the end point and kin_out):
 * bk_dr_level_begin for the produced lanes (H = -((-logp) + kin), h = 0, live = 1; drghmc.py:421 -> :249-251);
 * lanes_out (may be NULL) receives the number of lanes worked on, lanes_total (may be NULL) is incremented
 * by it (statistics: gradient evaluations = steps * lanes). */
int bk_leapfrog_finish_level(const double* rho_in, double* rho_out, int64_t ld, const double* grad,
                             int64_t ldg_d, int64_t ldg_c, const double* metric, double half, int negate,
                             double* kin_out, int64_t C, int64_t D, const uint32_t* n_dev,
                             const double* logp, double* H_out, double* h_out, uint8_t* live_out,
                             uint32_t* lanes_out, uint64_t* lanes_total, void* stream);

/* ---- Metropolis / Metropolis-Hastings accept ------------------------------------------
 * mode BK_ACCEPT_HMC  (hmc.py:57-63):  h0 = lp_cur - a_cur ; h1 = lp_prop - a_prop ;
 *        accept = log_u < h1 - h0 ;  ret[c] = accept ? h1 : h0   (joint log density)
 * mode BK_ACCEPT_MALA (mala.py:50-61, metropolis.py:70-76): a_cur = lp_forward,
 *        a_prop = lp_reverse ; accept = log_u < (lp_prop - lp_cur) + (a_prop - a_cur) ;
 *        ret[c] = accept ? lp_prop : lp_cur   (model log density, mala.py:66)
 * On accept lp_cur[c] <- lp_prop[c].  accept_mask[c] = 0/1.  *accept_count (device u32,
 * may be NULL) is incremented by the number of accepted chains, one atomic per wavefront
 * (ballot + popcount).  Strict `<` as in the reference. */
#define BK_ACCEPT_HMC 0
#define BK_ACCEPT_MALA 1
int bk_mh_accept(int mode, double* lp_cur, const double* a_cur, const double* lp_prop,
                 const double* a_prop, const double* log_u, uint8_t* accept_mask,
                 double* ret, uint32_t* accept_count, int64_t C, void* stream);

/* dst[d*ld + c] = mask[c] ? src[d*ld + c] : dst[d*ld + c] for up to two array pairs
 * (pair 1 may be NULL): `self._theta = theta_prop` (hmc.py:61, mala.py:62-64,
 * drghmc.py:379) applied to the accepted chains only.  Rejected chains keep their values; on
 * the 16-byte path (even C and ld, aligned pointers) they are re-written with them, so that no
 * partially written HBM sector results (a blend streams at full rate, holes do not).
 * `copy0` (may be NULL) additionally receives array 0 as it stands after the select, for every
 * chain (the stable array that sample() returns while the sampler keeps mutating its own; same
 * leading dimension). */
int bk_select_columns(const uint8_t* mask, double* dst0, const double* src0,
                      double* dst1, const double* src1, double* copy0, int64_t ld,
                      int64_t C, int64_t D, void* stream);

/* out[d*ld + c] = mask[c] ? b[d*ld + c] : a[d*ld + c]; a and b are only read.  The select for a sampler
 * that rebinds its state array every draw, as the reference does (`self._theta = theta_prop`,
 * hmc.py:61): `out` is the new state AND the array sample() returns, never written again.  About
 * 17 bytes of HBM traffic per element against 25 for bk_select_columns with copy0. */
int bk_blend_columns(const uint8_t* mask, const double* a, const double* b, double* out,
                     int64_t ld, int64_t C, int64_t D, void* stream);

/* ---- delayed rejection (DRGHMC) stage helpers -------------------------------------------
 * The reference's recursive accept() with its gradient-cache stack (drghmc.py:82,391-446)
 * is run as a lockstep state machine over lane sets: the chains still inside the stage
 * loop are compacted into dense buffers per recursion level, so a trajectory only costs
 * bandwidth for the chains that actually take it.  Per-chain scalars of a level: H (joint
 * log density), h (log of the probability of rejecting all earlier proposals, the
 * "hastings" term), a (log acceptance probability), live (not yet early-exited).
 *
 * Lane counts on the device.  Every entry point that works on a lane set takes, besides the
 * host-side extent `n` (or `m`), an optional `n_dev`: a device pointer to the number of lanes
 * actually in the set, as written by bk_compact_indices earlier on the same stream.  With
 * n_dev != NULL the launch is sized for `n` (an upper bound, e.g. the parent set) and works on
 * min(n, *n_dev) lanes; surplus workgroups exit at once.  The host then never reads a count
 * back, and a whole delayed-rejection draw -- whose lane sets depend on the draw's own accept
 * decisions -- is a fixed launch sequence that can be captured as one hipGraph.
 */

/* Stable compaction: idx_out[k] = position of the k-th nonzero entry of mask[0..n),
 * *count_out = number of nonzero entries (ballot/popcount prefix sums per wavefront). */
int bk_compact_indices(const uint8_t* mask, int64_t n, int32_t* idx_out, uint32_t* count_out,
                       const uint32_t* n_dev, void* stream);

/* Start of a draw (drghmc.py:365-366): cur_H = -((-logp) + kin) (joint_logp, :249-251),
 * cur_h = 0, rej = 0, alive = 1. */
int bk_dr_begin(const double* logp, const double* kin, double* cur_H, double* cur_h, double* rej,
                uint8_t* alive, int64_t C, void* stream);

/* Retry test (drghmc.py:369-371) for the alive chains: u from the chain's stream, chain
 * leaves the stage loop unless log(u) < prob_retry * rej  (prob_retry is 1.0 or 0.0; the
 * product reproduces False * -inf = nan -> break). */
int bk_dr_retry_test(int rng_kind, uint64_t* state, int64_t ldr, const double* rej,
                     double prob_retry, uint8_t* alive, int64_t C, void* stream);

/* Level set-up for accept(): H = -((-logp) + kin) (drghmc.py:421 -> :249-251), h = 0,
 * live = 1 for the n lanes of a level. */
int bk_dr_level_begin(const double* logp, const double* kin, double* H, double* h, uint8_t* live,
                      int64_t n, const uint32_t* n_dev, void* stream);

/* After the recursive accept of ghost proposals (drghmc.py:426-436): ghost lane j belongs
 * to parent lane p = sub_index ? sub_index[j] : j.  If ga[j] == 0 the parent's result is
 * a = -inf and it stops (early-out, :430-432); otherwise h[p] += log1p(-exp(ga[j])). */
int bk_dr_ghost_update(const double* ga, const int32_t* sub_index, int64_t m, double* h,
                       uint8_t* live, double* a, const uint32_t* n_dev, void* stream);

/* Final acceptance probability of the still-live lanes (drghmc.py:438-446):
 * a = min(0, (H - cur_H) + (h - cur_h) + (pr*h - pr*cur_h)), with cur_* read at
 * cur_index ? cur_index[j] : j. */
int bk_dr_accept_prob(const double* H, const double* cur_H, const double* h, const double* cur_h,
                      const int32_t* cur_index, double prob_retry, const uint8_t* live, double* a,
                      int64_t n, const uint32_t* n_dev, void* stream);

/* Top-level accept test of a stage (drghmc.py:378-385) for the n compacted lanes; lane j is
 * chain g = chain_index ? chain_index[j] : j.  u from chain g's stream; if log(u) < a[j]:
 * accepted[j] = 1, cur_H[g] = H[j], alive[g] = 0; else accepted[j] = 0,
 * rej[g] = log1p(-exp(a[j])), cur_h[g] += rej[g]. */
int bk_dr_accept_test(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index,
                      const double* a, const double* H, int64_t n, double* cur_H, double* cur_h,
                      double* rej, uint8_t* alive, uint8_t* accepted, const uint32_t* n_dev,
                      void* stream);

/* bk_dr_accept_prob for the chains' CURRENT point followed by bk_dr_accept_test, in one launch: lane j's
 * accept probability a[j] (written, if live[j]) is evaluated against cur_H / cur_h of its chain right before
 * the test draws its uniform.  Same arithmetic, one launch less per stage. */
int bk_dr_accept_prob_test(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index,
                           const double* H, const double* h, const uint8_t* live, double* a,
                           double prob_retry, int64_t n, double* cur_H, double* cur_h, double* rej,
                           uint8_t* alive, uint8_t* accepted, const uint32_t* n_dev, void* stream);

/* Start of a draw and the first stage's retry test in one launch: bk_dr_begin, then bk_dr_retry_test for every
 * chain (drghmc.py:365-371: reject_logp = 0 at the first stage, the test passes, its uniform is drawn all the
 * same).  Also zeroes `n_counters` (<= 64) lane counters -- the `next_count` words of the two entry points below --
 * so that a whole draw needs no separate memset -- and adds 1 to *draw_counter (may be NULL): the sampler's draw
 * count in device memory, which bk_welford_update_dev / bk_record_series_dev launched later in the same draw read. */
int bk_dr_begin_retry(int rng_kind, uint64_t* state, int64_t ldr, const double* logp, const double* kin,
                      double* cur_H, double* cur_h, double* rej, uint8_t* alive, double prob_retry,
                      uint32_t* counters, int64_t n_counters, int64_t* draw_counter, int64_t C, void* stream);

/* bk_dr_accept_prob_test, then -- for the lanes it rejects -- the NEXT stage's retry test (bk_dr_retry_test: the
 * chain's next uniform, drghmc.py:369-371) and the compaction of the chains that propose again: chain g is
 * appended to next_index[0 .. *next_count) (one atomic per wavefront; *next_count must be zero beforehand).  The
 * list holds the same chains bk_compact_indices would give, in an order that depends on wavefront timing: lane
 * order carries no meaning (every chain's values are independent of the lane that computes them).
 * next_index must not be chain_index.  drghmc.py:378-385 and :369-371. */
int bk_dr_accept_prob_test_next(int rng_kind, uint64_t* state, int64_t ldr, const int32_t* chain_index,
                                const double* H, const double* h, const uint8_t* live, double* a,
                                double prob_retry, int64_t n, double* cur_H, double* cur_h, double* rej,
                                uint8_t* alive, uint8_t* accepted, const uint32_t* n_dev, int32_t* next_index,
                                uint32_t* next_count, void* stream);

/* bk_dr_accept_prob_ghost, then the parent lanes that are still live -- the lane set of the parent's NEXT ghost
 * (drghmc.py:424) -- are appended to next_index[0 .. *next_count) as above (instead of a bk_compact_indices
 * launch over parent_live).  next_index must not be sub_index. */
int bk_dr_accept_prob_ghost_next(const double* H, const double* parent_H, const double* h, double* parent_h,
                                 const int32_t* sub_index, double prob_retry, const uint8_t* live, double* a,
                                 int64_t n, const uint32_t* n_dev, uint8_t* parent_live, double* parent_a,
                                 int32_t* next_index, uint32_t* next_count, void* stream);

/* bk_dr_accept_prob of a GHOST level (against its parent level's H / h, lanes paired by sub_index) followed
 * by bk_dr_ghost_update of the parent (parent_h, parent_live, parent_a), in one launch: every ghost lane has
 * exactly one parent lane.  drghmc.py:426-446. */
int bk_dr_accept_prob_ghost(const double* H, const double* parent_H, const double* h, double* parent_h,
                            const int32_t* sub_index, double prob_retry, const uint8_t* live, double* a,
                            int64_t n, const uint32_t* n_dev, uint8_t* parent_live, double* parent_a,
                            void* stream);

/* Accepted lanes replace their chain's current point (drghmc.py:379): for up to three
 * array pairs dst[d*ld_dst + g] = src[d*ld_src + j] and one per-chain vector
 * sdst[g] = ssrc[j], where g = index ? index[j] : j and mask[j] != 0. */
int bk_scatter_columns(const uint8_t* mask, const int32_t* index, int64_t n, int64_t D,
                       double* dst0, const double* src0, double* dst1, const double* src1,
                       double* dst2, const double* src2, int64_t ld_dst, int64_t ld_src,
                       double* sdst, const double* ssrc, const uint32_t* n_dev, void* stream);

/* ---- MALA ---------------------------------------------------------------------------
 * theta_prop = (theta + eps*grad) + sqrt2eps * z, z from chain c's stream in d order
 * (mala.py:41-45). */
int bk_mala_propose(int rng_kind, uint64_t* state, int64_t ldr, const double* theta,
                    const double* grad, double* theta_prop, int64_t ld, double eps,
                    double sqrt2eps, int64_t C, int64_t D, void* stream);

/* The same proposal with the D normals already drawn (e.g. generated ahead on another stream):
 * theta_prop = (theta + eps*grad) + sqrt2eps * z, with z[d*z_stride_d + c*z_stride_c] either in
 * the state layout (strides ld, 1; from bk_momentum_refresh) or chain-major (strides 1, ldz;
 * from bk_normals_chain_major, turned through LDS tiles here). */
int bk_mala_propose_from_normals(const double* theta, const double* grad, const double* z,
                                 int64_t z_stride_d, int64_t z_stride_c, double* theta_prop,
                                 int64_t ld, double eps, double sqrt2eps, int64_t C, int64_t D,
                                 void* stream);

/* zt[c*ldz + d] = d-th next standard normal of chain c's stream, d = 0..D-1 (what
 * `rng.normal(size=D)` returns, mala.py:44 / hmc.py:56), 16 to 64 lanes of a wavefront per chain,
 * Philox streams only; state advanced exactly as by sequential consumption.  ldz >= D.
 * `snapshot` (may be NULL; a second table with the same ldr) receives the stream table as it was BEFORE
 * the call: a sampler that generates a draw's normals one draw ahead keeps its logical stream
 * position (where the reference's generator stands between two sample() calls) that way, without a
 * separate copy of the table.
 * max_workgroups (0 = no bound): a BACKGROUND launch of at most that many workgroups, each walking several groups of
 * chains -- with one workgroup per CU (256 on MI355X) the generator keeps one wavefront per SIMD and the rest of every CU
 * stays free for a bandwidth-bound kernel running beside it on another stream.  Same stream of normals. */
int bk_normals_chain_major(int rng_kind, uint64_t* state, int64_t ldr, double* zt, int64_t ldz, int64_t C,
                           int64_t D, uint64_t* snapshot, int64_t max_workgroups, void* stream);

/* ONE chain driven by a host model (the reference's own call shape, README.md:13-32; mala.py:40-66): everything of a
 * draw that follows the model call, and the next draw's proposal, in ONE launch of one lane.
 *   have_prop != 0: host_in[0] = log density at theta_prop, host_in[1 .. D] = its gradient (written by the caller
 *     into host memory the device can address: pinned + mapped); proposal densities as bk_mala_logq, accept test
 *     as bk_mh_accept(BK_ACCEPT_MALA) with the next uniform of the chain's stream, select into theta / grad / lp
 *     (device, [D], [D], [1]); host_out[0 .. D) = theta after the draw, host_out[D] = the returned log density
 *     (mala.py:66), host_out[D + 1] = 1.0 / 0.0 accepted.
 *   always: snapshot (may be NULL; a table like state) = the stream table at this point -- where the reference's
 *     generator stands between two sample() calls --; then theta_prop = (theta + eps*grad) + sqrt2eps*z from the next
 *     D normals (mala.py:41-45), also to host_out[D + 2 .. 2D + 2); finally host_out[2D + 2] = seq, written last
 *     (system-scope release) so that the host may wait on it instead of synchronising the stream.
 * state: the table of ONE chain (column 0 of a table with row pitch ldr). */
int bk_mala_single_draw(int rng_kind, uint64_t* state, int64_t ldr, double* theta, double* grad, double* lp,
                        double* theta_prop, const double* host_in, double* host_out, uint64_t* snapshot,
                        int64_t lds, uint8_t* accept_mask, uint32_t* accept_count, double eps, double sqrt2eps,
                        int64_t D, int have_prop, double seq, void* stream);

/* lp_forward[c] = (-0.25/eps) * |(theta_prop - theta) - eps*grad|^2       mala.py:50-52
 * lp_reverse[c] = (-0.25/eps) * |(theta - theta_prop) - eps*grad_prop|^2   mala.py:53,68-79 */
int bk_mala_logq(const double* theta, const double* grad, const double* theta_prop,
                 const double* grad_prop, int64_t ld, double eps, double* lp_forward,
                 double* lp_reverse, int64_t C, int64_t D, void* stream);

/* The rest of a MALA draw in ONE pass over HBM (mala.py:50-66), plus the next draw's proposal
 * (mala.py:41-45).  A workgroup owns 16 chains x all D dimensions and keeps them on chip (three
 * arrays in registers, grad_prop in LDS) between the per-chain sums and the select:
 *     fwd, rev as bk_mala_logq                                   mala.py:50-53, 68-79
 *     accept[c] = log_u[c] < (lp_prop - lp) + (rev - fwd)        metropolis.py:70-76 (strict <)
 *     theta_out = accept ? theta_prop : theta                    mala.py:62   (theta_out may be theta)
 *     grad      = accept ? grad_prop  : grad      (in place)     mala.py:64
 *     lp        = accept ? lp_prop    : lp        (in place);  ret[c] = the same   mala.py:63, :66
 *     accept_mask[c], *accept_count += number accepted            (each may be NULL)
 * and, if zt_next != NULL (chain-major normals zt_next[c*ldz + d] of the NEXT draw, from
 * bk_normals_chain_major):
 *     theta_prop = (theta_out + eps*grad) + sqrt2eps * z          mala.py:41-45   (in place)
 * Every element of theta_out / grad is rewritten (rejected chains with their own values: whole
 * sectors, no read-modify-write in the memory system).  HBM traffic: 32*D read + 16*D written per
 * chain, + 8*D read and 8*D written with zt_next -- with the model's gradient op (16*D) and the
 * generator's write of zt (8*D) a draw moves 88*D bytes per chain.
 * The per-chain sums run per thread over rows r, r+64, ..., then over a fixed xor tree of the 8
 * rows of a wavefront, then over the 8 wavefronts in order: an order that depends on D only.
 * Supported shapes (bk_mala_step_supported): D <= 1024, C and ld even, 16-byte aligned arrays;
 * otherwise BK_E_ALIGN -- callers then use bk_mala_logq + bk_mh_accept + bk_select_columns. */
int bk_mala_step_supported(int64_t C, int64_t D, int64_t ld);
int bk_mala_step(const double* theta, double* theta_out, double* grad, double* theta_prop,
                 const double* grad_prop, int64_t ld, double* lp, const double* lp_prop,
                 const double* log_u, const double* zt_next, int64_t ldz, double eps,
                 double sqrt2eps, uint8_t* accept_mask, double* ret, uint32_t* accept_count,
                 int64_t C, int64_t D, void* stream);

/* ---- built-in targets: the "thin C-ABI callback" form of GradModel.log_density_gradient
 * (typing.py:25-27) batched over chains.  grad and/or logp may be NULL (HMC discards lp
 * inside the trajectory, hmc.py:45,50).  Operation order = oracle/models.py.
 *   iso     : logp = -0.5*sum th^2            grad = -th
 *   diag    : t = lam*th; logp = -0.5*sum th*t; grad = -t
 *   funnel  : v = th[0], n = D-1, ev = exp(-v), s = sum_{i>=1} th_i^2 in the library's canonical order for
 *             lane-spread densities (csrc/bk_lanes.hpp), for every D: row d is in class c = (d-1) mod 16;
 *             cs[c] = the class's rows summed in order; q[g] = ((cs[g] + cs[g+4]) + cs[g+8]) + cs[g+12];
 *             s = ((q[0] + q[1]) + q[2]) + q[3]
 *             logp = ((-(v*v)/18) - (0.5*n)*v) - (0.5*ev)*s
 *             grad0 = ((-v/9) - 0.5*n) + (0.5*ev)*s ; grad_i = -(ev*th_i)
 */
int bk_target_iso_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld,
                                int64_t C, int64_t D, void* stream);
int bk_target_diag_gaussian_grad(const double* theta, double* grad, double* logp, int64_t ld,
                                 const double* lam, int64_t C, int64_t D, void* stream);
int bk_target_funnel_grad(const double* theta, double* grad, double* logp, int64_t ld,
                          int64_t C, int64_t D, void* stream);

/* The same gradients for a lane set whose size lives on the device (n_dev, see bk_leapfrog_kick_drift_n):
 * chains [0, min(C, *n_dev)) of the arrays are evaluated, the launch is sized for C. */
int bk_target_iso_gaussian_grad_n(const double* theta, double* grad, double* logp, int64_t ld,
                                  int64_t C, int64_t D, const uint32_t* n_dev, void* stream);
int bk_target_diag_gaussian_grad_n(const double* theta, double* grad, double* logp, int64_t ld,
                                   const double* lam, int64_t C, int64_t D, const uint32_t* n_dev,
                                   void* stream);
int bk_target_funnel_grad_n(const double* theta, double* grad, double* logp, int64_t ld,
                            int64_t C, int64_t D, const uint32_t* n_dev, void* stream);

/* One leapfrog step {gradient, kick, drift} (drghmc.py:280-283; hmc.py:48-50) of Neal's funnel as ONE launch:
 * rho += h * (metric * grad(theta)); theta += h * rho, in place over chains [0, min(n, *n_dev)) (n_dev may be NULL).
 * The same arithmetic as bk_target_funnel_grad_n followed by bk_leapfrog_kick_drift_n -- bit-identical -- in half the
 * launches; the gradient never travels through memory.  (An instantiation of csrc/bk_lanes.hpp's k_lane_op, like the
 * funnel's gradient op; CTarget.from_source(form="lanes") exports the same entry for a user density.) */
int bk_leapfrog_step_funnel(double* theta, double* rho, int64_t ld, const double* metric, double h, int64_t n,
                            int64_t D, const uint32_t* n_dev, void* stream);

/* A whole HMC trajectory (hmc.py:40-53: backward half-kick, steps x {kick, drift, gradient}, forward half-kick) of every
 * chain on Neal's funnel in ONE launch, the gradient inlined (an instantiation of csrc/bk_lanes.hpp's trajectory kernel -- the
 * one behind bk_dr_proposal_funnel -- with hmc.py's first kick; requires D - 1 <= 128 and steps >= 1, BK_E_ARG otherwise).
 * In: theta_in, the momentum rho (OVERWRITTEN with minus the end momentum, which HMC discards) and grad_in, the gradient at
 * theta_in; out: theta_out, grad_out / logp_out there, kin_out = 1/2 sum rho*(metric*rho) at the end (hmc.py:59 -> :37).
 * All [D][ld] arrays share ld.  Follow with bk_mh_accept(BK_ACCEPT_HMC, ...) and bk_select_columns (theta AND grad).
 * CTarget.from_source(form="lanes") exports the same entry for a user density (bk_src_hmc_trajectory_lanes). */
int bk_hmc_trajectory_funnel(const double* theta_in, double* rho, const double* grad_in, double* theta_out,
                             double* grad_out, double* logp_out, double* kin_out, int64_t ld,
                             const double* metric, double eps, int64_t steps, int64_t C, int64_t D, void* stream);

/* ---- user targets (plugin ABI) ------------------------------------------------------------
 * A model the USER compiles into their own shared library plugs in below the samplers through
 * one exported function of this type: GradModel.log_density_gradient (typing.py:25-27) for all
 * chains at once.  theta[d*ld + c] in; grad[d*ld + c] and logp[c] out (either may be NULL);
 * `params` is an opaque pointer owned by the model (host or device memory, its choice); the
 * function only enqueues on `stream` and returns 0 or a nonzero status (hipError_t or its own
 * negative codes).  bayes_kit_amd.CTarget(library, symbol, dims, params) loads it; the samplers
 * then treat it exactly like the bk_target_* functions above (no torch, no Python in the
 * gradient call).  examples/plugin_target/ar1_target.hip is a complete one. */
typedef int (*bk_target_fn)(const double* theta, double* grad, double* logp, int64_t ld,
                            const void* params, int64_t C, int64_t D, void* stream);

/* The counted form of the plugin ABI (optional second export of a user target): as bk_target_fn, with the
 * number of chains to evaluate read ON THE DEVICE -- chains [0, min(C, *n_dev)), n_dev never NULL, the launch
 * sized for the bound C, surplus workgroups exit (one scalar load + compare per workgroup).  A target that
 * exports it (bayes_kit_amd.CTarget(..., counted_symbol=...)) runs DrGhmcDiag without any host
 * synchronisation inside a draw: the gradient calls of drghmc.py:280-283 for the data-dependent lane sets of
 * a delayed-rejection stage become part of one captured hipGraph.  examples/plugin_target/funnel_target.hip
 * exports both forms. */
typedef int (*bk_target_fn_n)(const double* theta, double* grad, double* logp, int64_t ld,
                              const void* params, int64_t C, int64_t D, const uint32_t* n_dev,
                              void* stream);

/* Whole HMC trajectory (hmc.py:40-53) for the separable Gaussian targets with the gradient
 * callback inlined: back half-step, `steps` x (kick, drift, grad = -(lam*theta)), forward
 * half-step, all in registers.  lam NULL = iso Gaussian; metric NULL = ones.  Outputs may
 * alias inputs.  Bit-identical to `steps` calls of bk_leapfrog_kick_drift +
 * bk_target_*_gaussian_grad followed by bk_leapfrog_finish's half-kick; HBM traffic 32 B per
 * element per trajectory; bound by the fp64 vector rate (6 flop per element-step). */
int bk_hmc_trajectory_gaussian(const double* theta_in, double* theta_out, const double* rho_in,
                               double* rho_out, int64_t ld, const double* lam, const double* metric,
                               double eps, int64_t steps, int64_t C, int64_t D, void* stream);

/* Trajectory AND energies of one HMC draw (hmc.py:55-59) on the separable Gaussian targets, in
 * one pass over the state: the trajectory above, plus
 *     kin0[c]   = 0.5 * sum_d rho0*(metric*rho0)     hmc.py:57 -> :37   (optional, may be NULL)
 *     kin1[c]   = 0.5 * sum_d rho1*(metric*rho1)     hmc.py:59 -> :37
 *     lp_out[c] = -0.5 * sum_d theta1*(lam*theta1)   the target's log density at the end point
 * with the per-chain sums taken exactly as bk_leapfrog_finish / bk_target_*_gaussian_grad take
 * them (four contiguous quarters of the dimensions, each sequential in d, combined
 * ((p0+p1)+p2)+p3), so the accept test sees bit-identical energies.  The momentum comes either
 * in the state layout (rho_in [D][ld]) or chain-major straight from bk_normals_chain_major
 * (zt[c*ldz + d], rho0 = 0.0 + 1.0*z: `rng.normal(size=D)`, hmc.py:56); exactly one of the two is
 * given.  The end momentum is not stored (hmc.py:58-63 never uses it again).
 * part: caller scratch of 12*C doubles.  HBM traffic 24*D bytes per chain.
 * With lp_cur != NULL (the target's log density at theta_in, per chain) and log_u, the accept test
 * of the draw (hmc.py:60-63) is evaluated by the same launch sequence, exactly as
 * bk_mh_accept(BK_ACCEPT_HMC, lp_cur, kin0, lp_out, kin1, log_u, accept_mask, ret, accept_count)
 * would: accept_mask / ret / accept_count optional, lp_cur updated in place on accepted chains. */
int bk_hmc_draw_gaussian(const double* theta_in, double* theta_out, int64_t ld, const double* rho_in,
                         const double* zt, int64_t ldz, const double* lam, const double* metric,
                         double eps, int64_t steps, double* part, double* kin0, double* kin1,
                         double* lp_out, double* lp_cur, const double* log_u, uint8_t* accept_mask,
                         double* ret, uint32_t* accept_count, int64_t C, int64_t D, void* stream);

/* bk_mala_step for a SEPARABLE built-in density (the Gaussians: lam NULL = identity): proposal densities, accept, select and
 * the next draw's proposal (bayes_kit/mala.py:41-66) with both gradients RECOMPUTED from theta / theta_prop where bk_mala_step
 * loads them, and the gradient at the new state never stored -- 56*D bytes per chain-draw with the log-density launch
 * (lp_prop = bk_target_*_gaussian_grad with grad NULL) instead of 88*D.  Same arithmetic on the same values as
 * {bk_target_*_gaussian_grad, bk_mala_step}: bit-identical draws.  Arguments as bk_mala_step without grad / grad_prop. */
int bk_mala_step_gaussian(const double* theta, double* theta_out, double* theta_prop, int64_t ld, const double* lam,
                          double* lp, const double* lp_prop, const double* log_u, const double* zt_next, int64_t ldz,
                          double eps, double sqrt2eps, uint8_t* accept_mask, double* ret, uint32_t* accept_count, int64_t C,
                          int64_t D, void* stream);

/* One whole delayed-rejection proposal (drghmc.py:319-346 -> :253-289) on Neal's funnel in a
 * single launch, gradient callback inlined: chain j of the outputs starts from chain
 * src_index[j] (NULL = j) of the source point (theta_in, rho_in and the source's cached
 * gradient grad_in); first half-kick + drift, (steps-1) x {gradient, kick, drift}, final
 * gradient and log density, last half-kick, momentum flip.  Outputs (compact, ld_out):
 * theta, -rho, gradient and log density at the proposal, kin = 0.5 sum rho*(metric*rho).
 * Requires D - 1 <= 128 (returns BK_E_ARG otherwise: use the step-by-step entry points).
 * The sum over coordinates uses the same fixed order as bk_target_funnel_grad, so the
 * proposal is bit-identical to the step-by-step path.
 * n_dev (may be NULL): device-side lane count, see "Lane counts on the device" above.
 * lanes_out (may be NULL): receives the number of lanes the launch worked on (statistics:
 * gradient evaluations = steps * lanes); lanes_total (may be NULL): the same count is ADDED to it.
 * H_out, h_out, live_out (all or none): additionally perform bk_dr_level_begin for the produced
 * lanes in the same launch (H = -((-logp) + kin), h = 0, live = 1).
 * grad_in NULL: the launch evaluates the source point's gradient itself instead of reading a cached one (the same
 * values: the gradient is a function of theta alone and its sums have one order); grad_out NULL: the end point's gradient
 * is not stored.  A caller that keeps no gradient cache at all (DrGhmcDiag's one-launch path) moves two arrays per launch
 * and per scatter instead of three. */
/* (prototype below, after the structs its optional jobs are described by.) */

/* The arguments of one bk_scatter_columns call, as a job a trajectory launch can carry along. */
typedef struct bk_scatter_job {
  const uint8_t* mask;
  const int32_t* index;
  int64_t n, D;
  double* dst0; const double* src0;
  double* dst1; const double* src1;
  double* dst2; const double* src2;
  int64_t ld_dst, ld_src;
  double* sdst; const double* ssrc;
  const uint32_t* n_dev;
} bk_scatter_job;

/* What a GHOST proposal without ghosts of its own (the first proposal kind: drghmc.py:424 with k = 0) owes its
 * parent level, so that the proposal's launch can do bk_dr_accept_prob_ghost[_next] itself: lane j's parent lane is
 * src_index[j] (the lane it was gathered from).  a_out: the ghost level's `a`; next_index / next_count may be NULL. */
typedef struct bk_ghost_link {
  const double* parent_H;
  double* parent_h;
  uint8_t* parent_live;
  double* parent_a;
  double* a_out;
  double prob_retry;
  int32_t* next_index;
  uint32_t* next_count;
} bk_ghost_link;

/* The FIRST GHOST of the produced proposals (drghmc.py:424 with i = 0), integrated by the proposal's own launch from its
 * registers: h / steps of the first proposal kind.  Every produced lane gets one, lane for lane; only the ghost's joint
 * log density is used, so no ghost arrays exist.  The launch then writes the level's h (= log1p(-exp(g)), or 0) and live
 * (g != 0) itself, parent_a[j] = -inf where g == 0, and appends the lanes that go on to their next ghost to
 * next_index (may be NULL).  lanes_out / lanes_total (may be NULL): the ghost trajectory's lane statistics. */
typedef struct bk_ghost0 {
  double h;
  int64_t steps;
  double* parent_a;
  double prob_retry;
  int32_t* next_index;
  uint32_t* next_count;
  uint32_t* lanes_out;
  uint64_t* lanes_total;
} bk_ghost0;

/* job (may be NULL): a scatter job run by surplus workgroups of the SAME launch: the
 * previous stage's accepted columns move into the chains' current point (drghmc.py:379-381) while this stage's
 * trajectories -- a sparse, latency-bound lane set -- integrate.  The caller guarantees that the job and the
 * proposal touch disjoint memory: the job writes columns of accepted chains and reads the previous stage's
 * proposal buffers, the proposal reads columns of rejected chains and writes its own buffers.
 * ghost (may be NULL; needs H_out): the launch also evaluates each produced lane's acceptance probability against
 * its parent lane and applies it to the parent (bk_ghost_link): one launch instead of two per such ghost.
 * ghost0 (may be NULL; needs H_out): the launch also integrates the first ghost of every produced lane (bk_ghost0) and
 * applies it to the produced level; together with `ghost` only for a level whose ONLY ghost it is
 * (ghost0->next_index NULL) -- the link then sees the level after that ghost. */
int bk_dr_proposal_funnel(const double* theta_in, const double* rho_in, const double* grad_in,
                          int64_t ld_in, const int32_t* src_index, double* theta_out,
                          double* rho_out, double* grad_out, double* logp_out, double* kin_out,
                          int64_t ld_out, const double* metric, double h, int64_t steps, int64_t n,
                          int64_t D, const uint32_t* n_dev, uint32_t* lanes_out, uint64_t* lanes_total,
                          double* H_out, double* h_out, uint8_t* live_out, const bk_scatter_job* job,
                              const bk_ghost_link* ghost, const bk_ghost0* ghost0, void* stream);

/* The separable Gaussians through the same kernel templates (a separable density is a lanes-form density without head
 * coordinates): bk_dr_proposal_funnel and bk_leapfrog_step_funnel for logp = -1/2 sum th*(lam*th), lam NULL = the
 * isotropic Gaussian.  D <= 128 for the proposal (BK_E_ARG otherwise), any D for the step.  theta and rho are bit-identical to
 * the step-by-step path; the log density is summed in the lanes' class order (csrc/bk_lanes.hpp), not in four quarters.
 * The step of a separable density needs no sums: bk_leapfrog_step_gaussian is a STREAMING launch (csrc/bk_elementwise.hpp,
 * every (d, c) element on its own, 16 bytes per lane): 32*D bytes per chain-step, min(n, *n_dev) chains. */
int bk_dr_proposal_gaussian(const double* theta_in, const double* rho_in, const double* grad_in,
                                int64_t ld_in, const int32_t* src_index, double* theta_out,
                                double* rho_out, double* grad_out, double* logp_out, double* kin_out,
                                int64_t ld_out, const double* metric, double h, int64_t steps, int64_t n,
                                int64_t D, const uint32_t* n_dev, uint32_t* lanes_out, uint64_t* lanes_total,
                                double* H_out, double* h_out, uint8_t* live_out, const bk_scatter_job* job,
                                const bk_ghost_link* ghost, const bk_ghost0* ghost0, const double* lam,
                                void* stream);
int bk_leapfrog_step_gaussian(double* theta, double* rho, int64_t ld, const double* lam, const double* metric,
                              double h, int64_t n, int64_t D, const uint32_t* n_dev, void* stream);

/* ---- dense mass matrix (no reference counterpart: parity unpinned) ----------------------------
 * Y[d*ld + c] = sum_k M[d*ldm + k] * X[k*ld + c] for all chains: one fp64 GEMM on the matrix
 * cores (v_mfma_f64_16x16x4_f64), 128 x 128 workgroup tiles, XCD-aware placement.  Used for
 * rho = chol(M) @ z, the kick's M @ grad and the kinetic energy's M^-1 @ rho, where the
 * reference has the elementwise `metric * grad` / `metric * rho` (hmc.py:37,46-52).
 * 2*D*D*C flop. */
int bk_dense_metric_apply(const double* M, int64_t ldm, const double* X, double* Y, int64_t ld,
                          int64_t C, int64_t D, void* stream);

/* The same MFMA GEMM with a rectangular left factor: Y[R x C] = A[R x K] @ X[K x C], all three
 * with the chain index contiguous on the right (design matrices: X_data @ Theta with R = number
 * of observations, X_data^T @ residuals with R = dims).  `work` (may be NULL; work_elems
 * doubles, caller-owned, at least bk_gemm_chains_work_elems(R, K, C) of them) lets the library split a
 * long inner dimension over several workgroups when the output has too few tiles to fill the chip; the
 * slabs are summed in a fixed order.  The split is a function of (R, K) only and calls wider than 2,048
 * chains are cut into column blocks: a chain's result does not depend on how many chains share its call
 * (chains sharded over GPUs reproduce the unsharded run bit for bit).  A `work` that is too small is
 * refused (BK_E_ARG), never answered with another split. */
int64_t bk_gemm_chains_work_elems(int64_t R, int64_t K, int64_t C);
int bk_gemm_chains(const double* A, int64_t lda, int64_t R, int64_t K, const double* X, int64_t ldx,
                   double* Y, int64_t ldy, int64_t C, double* work, int64_t work_elems, void* stream);

/* bk_gemm_chains whose result leaves as y_rows[r] - sigmoid(Y[r][c]): the logistic regression's first GEMM with the
 * residual pass (bk_logistic_residual without the log likelihood) in its epilogue -- a gradient-only evaluation is
 * then two launches.  Same values as the GEMM followed by bk_logistic_residual(part = NULL). */
int bk_gemm_chains_logistic(const double* A, int64_t lda, int64_t R, int64_t K, const double* X, int64_t ldx,
                            double* Y, int64_t ldy, int64_t C, const double* y_rows, void* stream);

/* Logistic regression target (BASELINE.json config 5; no reference oracle), between the two
 * GEMMs Z = X_data @ Theta and G = X_data^T @ R:
 *   bk_logistic_residual: Z[n*ldz + c] <- y[n] - sigmoid(z) in place, and
 *       part[s*C + c] = sum over the s-th block of observations of  y z - log(1 + e^z)
 *       (part NULL: gradient only -- no log likelihood is formed; bk_logistic_finish then takes part NULL too
 *       and writes grad only);
 *   bk_logistic_finish:   grad = t*G + (-(inv_prior_var*theta)),
 *       loglik[c] = sum_s part[s*C + c],  logp[c] = t*loglik + (-0.5*inv_prior_var*|theta|^2)
 *       (t = likelihood temperature of smc.py:47-51; 1 for plain sampling). */
int bk_logistic_residual(double* Z, int64_t ldz, const double* y, double* part, int64_t N, int64_t C,
                         int64_t segments, void* stream);
int bk_logistic_finish(const double* G, const double* theta, int64_t ld, const double* part,
                       int64_t segments, double inv_prior_var, double t, double* grad, double* logp,
                       double* loglik, int64_t C, int64_t D, void* stream);

/* out[c] = scale * sum_d x[d*ld + c] * y[d*ld + c]  (kinetic energy 0.5 * rho . (M rho)). */
int bk_dot_columns(const double* x, const double* y, int64_t ld, double scale, double* out,
                   int64_t C, int64_t D, void* stream);

/* ---- sequential Monte Carlo resampling (smc.py:64-75) ---------------------------------------
 * Multinomial resampling with exactly the arithmetic of the reference's
 *     np.random.choice(M, size=m, replace=True, p=weights / weights.sum())        (smc.py:73)
 * from the weights on: total = np.sum(weights) (numpy's pairwise summation, pieces of 8,192 values added in order),
 * p = weights / total, cdf = np.cumsum(p) as ONE sequential chain (written to cdf_work[n]),
 * idx_out[j] = searchsorted(cdf / cdf[n-1], u[j], side="right").  Given the uniforms RandomState.choice would
 * draw, the indices are bit-identical to the reference's (tests/golden/smc_*.npz). */
int bk_resample_indices(const double* weights, int64_t n, const double* u, int64_t m,
                        double* cdf_work, int32_t* idx_out, void* stream);

/* dst[d*ld_dst + j] = src[d*ld_src + index[j]]: the resampled particles (thetas[idxs]). */
int bk_gather_columns(const int32_t* index, const double* src, int64_t ld_src, double* dst,
                      int64_t ld_dst, int64_t m, int64_t D, void* stream);

/* ---- layout helper --------------------------------------------------------------------
 * dst[d*ld + c] = src[c*lds_c + d*lds_d]  (LDS-tiled transpose/copy) -- brings a model's
 * (C, D) row-major output into the engine's chain-contiguous layout, and back with the
 * roles of the strides swapped. */
int bk_relayout(const double* src, int64_t lds_d, int64_t lds_c, double* dst, int64_t ldd_d,
                int64_t ldd_c, int64_t C, int64_t D, void* stream);

/* ---- streaming diagnostics --------------------------------------------------------------
 * Welford update with the n-th draw (n >= 1) of every chain and dimension: per-chain mean
 * and M2 from which rhat.py:163-166 (np.mean, np.var(ddof=1)) follow. */
/* theta has a row pitch of its own (a sampler's state rows may be padded off a power-of-two pitch while the moments are
 * dense: no staging copy of the draw). */
int bk_welford_update(double* mean, double* m2, int64_t ld, const double* theta, int64_t ld_theta,
                      int64_t n, int64_t C, int64_t D, void* stream);
/* ... with the update count in device memory: n = *n_dev - n_offset (e.g. a sampler's draw counter).  A launch like this
 * can be part of a captured draw (hipGraph): nothing of it changes from one replay to the next on the host side. */
int bk_welford_update_dev(double* mean, double* m2, int64_t ld, const double* theta, int64_t ld_theta,
                          const int64_t* n_dev, int64_t n_offset, int64_t C, int64_t D, void* stream);

/* Draw storage for ess / rhat post-processing: row `row` of the K (+1) tracked series,
 * series[k][row][c] = theta[dims[k]][c] (k < K) and series[K][row][c] = logp[c] (logp may be NULL); series is
 * [K (+1)][capacity][C], chain-contiguous -- the [N, C] layout bk_ess / bk_chain_mean_var consume.  dims: device. */
int bk_record_series(const double* theta, int64_t ld, const int32_t* dims, int64_t K, const double* logp,
                     double* series, int64_t capacity, int64_t row, int64_t C, void* stream);

/* The same with row = *row_dev - row_offset read on the device (rows outside [0, capacity) are not written). */
int bk_record_series_dev(const double* theta, int64_t ld, const int32_t* dims, int64_t K, const double* logp,
                         double* series, int64_t capacity, const int64_t* row_dev, int64_t row_offset, int64_t C,
                         void* stream);

/* Per-dimension partial sums over this rank's C chains for R-hat (rhat.py:163-171):
 * out[0*D + d] = sum_c mean ; out[1*D + d] = sum_c var_c (var_c = m2/(n-1)).
 * With `center` != NULL also out[2*D + d] = sum_c (mean - center[d])^2  (second pass of
 * np.var(means, ddof=1)).  Deterministic fixed-shape tree per dimension. */
int bk_rhat_partials(const double* mean, const double* m2, int64_t ld, int64_t n,
                     const double* center, double* out, int64_t C, int64_t D, void* stream);

/* Per-chain mean and ddof=1 variance of a stored series x[t*ld + c], t < len[c] (len NULL =
 * all N): the two list comprehensions of rhat.py:165-166 for ragged chains. */
int bk_chain_mean_var(const double* x, int64_t ld, const int32_t* len, int64_t N,
                      double* mean, double* var, int64_t C, void* stream);

/* Rank normalisation (rhat.py:62-108): out[i] = Phi^-1((rank[i] - 0.325) / (S - 0.25)), ranks as
 * doubles (1-based), S = total number of draws; Phi^-1 is Cephes ndtri, the function behind
 * scipy.stats.norm.ppf, evaluated in the same order. */
int bk_rank_normalize(const double* rank, double S, double* out, int64_t n, void* stream);

/* Building blocks of pooled ranks (rhat.py:27-59: every draw replaced by its rank among ALL draws
 * of ALL chains) without replicating the draws on every rank -- a sample sort:
 *   bk_sort_by_key    stable ascending sort of (key, payload) pairs (keys double, -0.0 < +0.0, NaNs
 *                     last; equal keys keep their input order).  `work`: caller scratch of
 *                     bk_sort_by_key_work_bytes(n) bytes.  In and out arrays must not overlap.
 *   bk_count_below    out[i] = number of sorted_keys[0..n) strictly below queries[i]  (bucket
 *                     boundaries for the splitters)
 *   bk_scatter_ranks  out[payload[j]] = base + (j + 1), j < n: the (1-based, as doubles) ranks of a
 *                     sorted run whose first element has `base` elements before it. */
int64_t bk_sort_by_key_work_bytes(int64_t n);
int bk_sort_by_key(const double* keys_in, double* keys_out, const int64_t* vals_in, int64_t* vals_out,
                   int64_t n, void* work, int64_t work_bytes, void* stream);
int bk_count_below(const double* sorted_keys, int64_t n, const double* queries, int64_t m, int64_t* out,
                   void* stream);
int bk_scatter_ranks(const int64_t* payload, int64_t n, double base, double* out, void* stream);

/* Autocorrelation at all lags 0..N-1 of each chain of a stored series, out[n*ldo + c]
 * (autocorr.py:6-33; same normalisation: / np.var(x) / N).  Direct summation, the series staged in
 * LDS (16 chains per workgroup, one wavefront per chain, one lane per lag: N steps per 64 lags). */
int bk_autocorr(const double* x, int64_t ld, int64_t N, double* out, int64_t ldo, int64_t C,
                void* stream);

/* The same by FFT, as the reference computes it (autocorr.py:23-33: zero-padded to S = 2**ceil(log2(2N-1)), |fft|^2,
 * inverse, / var / N), for chains too long for the LDS-staged direct sums: O(N log N) per chain.  Stockham radix-16
 * passes across the rows of the [N, C] layout, one lane per column pair (two real series per complex transform);
 * work: caller-owned scratch of bk_autocorr_fft_work_bytes(N, C) bytes, 16-byte aligned. */
int64_t bk_autocorr_fft_work_bytes(int64_t N, int64_t C);
int bk_autocorr_fft(const double* x, int64_t ld, int64_t N, double* out, int64_t ldo, int64_t C, void* work,
                    int64_t work_bytes, void* stream);

/* IAT / ESS of each chain from an autocorrelation array acor[n*ld + c], n < N (iat.py:46-135: the
 * Geyer scan alone) -- for autocorrelations obtained elsewhere (bk_autocorr_fft for very long chains). */
int bk_iat_from_acor(const double* acor, int64_t ld, int64_t N, int estimator, double* ess_out,
                     double* iat_out, int64_t C, void* stream);

/* out[c] = index one past the last pair (0,1), (2,3), ... of acor[n*ld + c] before the first pair
 * with a negative sum (iat.py:7-43, the truncation point of both IAT estimators). */
int bk_end_pos_pairs(const double* acor, int64_t ld, int64_t N, int64_t* out, int64_t C,
                     void* stream);

/* ESS of each chain of a stored series (ess.py:52-69 -> iat.py:95-135 -> autocorr.py:6-33):
 * autocorrelations by direct summation (same quantity the reference gets by FFT) from an LDS-staged
 * tile -- 64 lags per N steps per wavefront, only as many 64-lag blocks as the truncation needs --, Geyer
 * initial-positive truncation at the first even lag pair with negative sum, initial
 * monotone running-min sum, IAT = 2*sum - 1, ESS = N/IAT.  estimator 0 = IMSE (ess /
 * ess_imse), 1 = IPSE (ess_ipse).  iat_out may be NULL. */
int bk_ess(const double* x, int64_t ld, int64_t N, int estimator, double* ess_out,
           double* iat_out, int64_t C, void* stream);

/* ---- host-side self-test hooks (tests only; they run the SAME source as the kernels on
 * the host so the RNG can be checked against numpy without a GPU).  Host pointers. */
int bk_host_normals(int rng_kind, uint64_t* state_words /*[BK_RNG_WORDS]*/, double* out,
                    int64_t n);
int bk_host_uniforms(int rng_kind, uint64_t* state_words, double* out, int64_t n);
double bk_host_log1p(double x);
/* bk_exp of include/bkhip_math.h (the funnel's exp: a specified fma sequence, restated in oracle/rng.py), host build. */
double bk_host_exp(double x);

#ifdef __cplusplus
}
#endif
#endif /* BKHIP_H */
