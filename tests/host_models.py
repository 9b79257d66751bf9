"""Reference-style single-chain host models used as fixtures (user code from the sampler's
point of view).  ``Binomial`` restates, in this repo's own words, the conjugate beta-binomial
test model of the reference (test/models/binomial.py:11-74) INCLUDING its finite-difference
"gradient" (lp(theta) - lp(theta + e)) / e, i.e. minus the forward difference: gradients are
opaque model output and are reproduced as they are."""
import numpy as np
from scipy import stats
from scipy.special import expit, log1p


class Binomial:
    def __init__(self, alpha, beta, x, N):
        self.alpha, self.beta, self.x, self.N = alpha, beta, x, N

    def dims(self):
        return 1

    def log_prior(self, params_unc):
        p = expit(params_unc[0])
        return stats.beta.logpdf(p, self.alpha, self.beta) + (np.log(p) + log1p(-p))

    def log_likelihood(self, params_unc):
        return stats.binom.logpmf(self.x, self.N, expit(params_unc[0]))

    def log_density(self, params_unc):
        return self.log_likelihood(params_unc) + self.log_prior(params_unc)

    def log_density_gradient(self, params_unc):
        e = 0.000001
        lp = self.log_density(params_unc)
        return lp, np.array([(lp - self.log_density(params_unc + e)) / e])

    def posterior_mean(self):
        return stats.beta(self.alpha + self.x, self.beta + self.N - self.x).mean()


class Ar1:
    """NumPy statement of examples/plugin_target/ar1_target.hip (a USER plugin target), same
    operation order: r_d = theta_d - a*theta_{d-1}; logp = (-0.5/s2) * sum r_d^2 summed in d
    order; grad_d = -((r_d - a*r_{d+1}) * (1/s2))."""

    def __init__(self, D, a, s2):
        self.D, self.a, self.inv_s2 = D, a, 1.0 / s2

    def dims(self):
        return self.D

    def _r(self, theta):
        prev = np.concatenate([[0.0], theta[:-1]])
        return theta - self.a * prev

    def log_density(self, theta):
        ss = 0.0
        for v in self._r(theta):
            ss = ss + v * v
        return (-0.5 * self.inv_s2) * ss

    def log_density_gradient(self, theta):
        r = self._r(theta)
        nxt = np.concatenate([r[1:], [0.0]])
        g = np.empty(self.D)
        g[:-1] = -((r[:-1] - self.a * nxt[:-1]) * self.inv_s2)
        g[-1] = -(r[-1] * self.inv_s2)
        return self.log_density(theta), g
