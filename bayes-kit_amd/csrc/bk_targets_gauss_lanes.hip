// The separable Gaussians through the lane-spread kernel templates (bk_lanes.hpp): a separable density is a lanes-form density
// WITHOUT head coordinates -- log p = -1/2 times ONE canonical-order sum of theta*(lam*theta), row gradient -(lam*theta) -- so
// the library's one-launch delayed-rejection proposal kernel (D <= 128) and one-launch leapfrog step serve IsoGaussian /
// DiagGaussian as they serve the funnel and every CTarget.from_source density.  theta and rho equal the step-by-step path's bit
// for bit (the gradient is elementwise); the log density is the same terms summed in the lanes' class order instead of four
// contiguous quarters (bk_target_*_gaussian_grad): last-bit differences.  A translation unit of its own: 130 kernel variants.
#include "bk_common.hpp"
#include "bk_elementwise.hpp"
#include "bk_lanes.hpp"

namespace {

template <bool HL>
struct GaussLanes {
  static constexpr int HEAD = 0;
  template <class L>
  __device__ __forceinline__ static double eval(L& c, const double* lam) {
    // (the sum only makes up the VALUE: skipped where the caller discards it -- every step but a trajectory's last)
    const double s = c.wants_logp() ? c.sum([lam](double x, i64 d) { const double lt = HL ? lam[d] * x : x; return x * lt; }) : 0.0;
    c.grad([lam](double x, i64 d) { const double lt = HL ? lam[d] * x : x; return -lt; });
    return -0.5 * s;
  }
};

template <bool HL>
struct GaussStepTerm {
  __device__ __forceinline__ static void eval(double th, i64 d, const double* lam, double& term, double& grad) {
    const double lt = HL ? lam[d] * th : th;
    term = th * lt;
    grad = -lt;
  }
  __device__ __forceinline__ static double finish(double s) { return -0.5 * s; }
};

}  // namespace

extern "C" {

int bk_dr_proposal_gaussian(const double* theta_in, const double* rho_in, const double* grad_in, int64_t ld_in,
                                const int32_t* src_index, double* theta_out, double* rho_out, double* grad_out,
                                double* logp_out, double* kin_out, int64_t ld_out, const double* metric, double h,
                                int64_t steps, int64_t n, int64_t D, const uint32_t* n_dev, uint32_t* lanes_out,
                                uint64_t* lanes_total, double* H_out, double* h_out, uint8_t* live_out,
                                const bk_scatter_job* job, const bk_ghost_link* ghost, const bk_ghost0* ghost0,
                                const double* lam, void* stream) {
  if (lam)
    return bkl::dr_proposal_launch<GaussLanes<true>>(theta_in, rho_in, grad_in, ld_in, src_index, theta_out, rho_out, grad_out,
                                                     logp_out, kin_out, ld_out, metric, h, steps, n, D, n_dev, lanes_out,
                                                     lanes_total, H_out, h_out, live_out, job, ghost, ghost0, lam, stream);
  return bkl::dr_proposal_launch<GaussLanes<false>>(theta_in, rho_in, grad_in, ld_in, src_index, theta_out, rho_out, grad_out,
                                                    logp_out, kin_out, ld_out, metric, h, steps, n, D, n_dev, lanes_out,
                                                    lanes_total, H_out, h_out, live_out, job, ghost, ghost0, nullptr, stream);
}

// (the step of a separable density needs no sums: the streaming elementwise kernel of bk_elementwise.hpp, every (d, c) element
// on its own -- 32 D bytes per chain-step -- rather than the lanes template's workgroup-per-64-chains op)
int bk_leapfrog_step_gaussian(double* theta, double* rho, int64_t ld, const double* lam, const double* metric, double h,
                              int64_t n, int64_t D, const uint32_t* n_dev, void* stream) {
  if (lam) return bke::step_launch<GaussStepTerm<true>>(theta, rho, ld, lam, metric, h, n, D, n_dev, stream);
  return bke::step_launch<GaussStepTerm<false>>(theta, rho, ld, nullptr, metric, h, n, D, n_dev, stream);
}

}  // extern "C"
