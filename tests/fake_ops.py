"""CPU stand-in for ``bayes_kit_amd._lib.Ops`` -- TEST INFRASTRUCTURE ONLY.

It lets the host-side control flow of the samplers (buffer plumbing, model bridge, call
patterns, the delayed-rejection state machine) run in the build container, which has no GPU.
Each method restates, with NumPy on CPU tensors, what the corresponding C-ABI entry point
of include/bkhip.h is specified to compute; random numbers come from numpy Generators
rebuilt from / written back to the same per-chain state table the device uses.  Nothing in
the product imports this file; on the GPU box the samplers run on the HIP library.
"""
import numpy as np
import torch

from bayes_kit_amd import _lib
from bayes_kit_amd._engine import _bitgen_words, numpy_generator_from_words


def _np(t):
    return None if t is None else t.numpy()


class FakeOps:
    name = "fake-cpu"

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # -- RNG ------------------------------------------------------------------------------
    def _gen(self, kind, state, c):
        return numpy_generator_from_words(kind, state.numpy().view(np.uint64)[:, c])

    def _put(self, kind, state, c, gen):
        k2, w = _bitgen_words(gen.bit_generator)
        assert k2 == kind
        state.numpy().view(np.uint64)[:, c] = w

    def rng_init_philox(self, state, key0, chain_id0):
        w = state.numpy().view(np.uint64)
        w[:] = 0
        w[0, :] = np.uint64(key0)
        w[1, :] = np.arange(w.shape[1], dtype=np.uint64) + np.uint64(chain_id0)
        w[10, :] = 4

    def refresh_work(self, C, D):
        return None

    def momentum_refresh(self, kind, state, loc_in, loc_mul, scale, out, metric, kin_out, active=None, work=None):
        self._count("momentum_refresh")
        D, C = out.shape
        o, li, m = _np(out), _np(loc_in), _np(metric)
        for c in range(C):
            if active is not None and not active[c]:
                continue
            g = self._gen(kind, state, c)
            z = g.standard_normal(D)
            loc = li[:, c] * loc_mul if li is not None else 0.0
            v = loc + scale * z
            o[:, c] = v
            if kin_out is not None:
                mv = m * v if m is not None else v
                kin = 0.0
                for d in range(D):
                    kin = kin + v[d] * mv[d]
                kin_out[c] = 0.5 * kin
            self._put(kind, state, c, g)

    def log_uniform(self, kind, state, out, active=None):
        self._count("log_uniform")
        for c in range(out.shape[0]):
            if active is not None and not active[c]:
                continue
            g = self._gen(kind, state, c)
            with np.errstate(divide="ignore"):
                out[c] = float(np.log(g.uniform()))
            self._put(kind, state, c, g)

    def uniform(self, kind, state, out, active=None):
        for c in range(out.shape[0]):
            if active is not None and not active[c]:
                continue
            g = self._gen(kind, state, c)
            out[c] = float(g.uniform())
            self._put(kind, state, c, g)

    # -- integrator ---------------------------------------------------------------------------
    @staticmethod
    def _mt(metric, g):
        return g if metric is None else metric.numpy()[:, None] * g

    def kick_drift(self, theta_in, theta_out, rho_in, rho_out, grad, metric, eps, use_pre, pre,
                   use_kick, kick, n_dev=None):
        self._count("kick_drift")
        if n_dev is not None:  # device-side lane count: only the first n lanes exist
            n = self._lanes(theta_out.shape[1], n_dev)
            theta_in, theta_out, rho_in, rho_out, grad = (t[:, :n] for t in (theta_in, theta_out, rho_in, rho_out, grad))
        t = self._mt(metric, grad.numpy())
        r = rho_in.numpy().copy()
        if use_pre:
            r = r + pre * t
        if use_kick:
            r = r + kick * t
        th = theta_in.numpy() + eps * r
        rho_out.numpy()[...] = r
        theta_out.numpy()[...] = th

    def first_step_gather(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, metric, eps, pre,
                          n_dev=None):
        self._count("first_step_gather")
        n = self._lanes(theta_out.shape[1], n_dev)
        theta_out, rho_out = theta_out[:, :n], rho_out[:, :n]
        idx = np.arange(n) if src_index is None else src_index.numpy()[:n]
        t = self._mt(metric, grad_in.numpy()[:, idx])
        r = rho_in.numpy()[:, idx] + pre * t
        rho_out.numpy()[...] = r
        theta_out.numpy()[...] = theta_in.numpy()[:, idx] + eps * r

    def leapfrog_finish(self, rho_in, rho_out, grad, metric, half, negate, kin_out, n_dev=None, level=None,
                        lanes_out=None, lanes_total=None):
        self._count("leapfrog_finish")
        if n_dev is not None or level is not None or lanes_out is not None or lanes_total is not None:
            n = self._lanes(rho_in.shape[1], n_dev)
            if lanes_out is not None:
                lanes_out[0] = n
            if lanes_total is not None:
                lanes_total[0] += n
            cut = lambda t: None if t is None else (t[:, :n] if t.dim() == 2 else t[:n])
            self.leapfrog_finish(cut(rho_in), cut(rho_out), cut(grad), metric, half, negate, cut(kin_out))
            if level is not None:
                logp, H, h, live = level
                self.dr_level_begin(logp, kin_out, H, h, live, n)
            return
        if grad is None:
            v = rho_in.numpy().copy()
        else:
            v = rho_in.numpy() + half * self._mt(metric, grad.numpy())
        if negate:
            v = -v
        if rho_out is not None:
            rho_out.numpy()[...] = v
        if kin_out is not None:
            mv = self._mt(metric, v)
            kin = np.zeros(v.shape[1])
            for d in range(v.shape[0]):
                kin = kin + v[d] * mv[d]
            kin_out.numpy()[...] = 0.5 * kin

    def mh_accept(self, mode, lp_cur, a_cur, lp_prop, a_prop, log_u, mask, ret, count):
        self._count("mh_accept")
        l0, l1 = lp_cur.numpy().copy(), lp_prop.numpy()
        a0 = a_cur.numpy() if a_cur is not None else 0.0
        a1 = a_prop.numpy() if a_prop is not None else 0.0
        with np.errstate(invalid="ignore"):
            if mode == _lib.ACCEPT_HMC:
                h0, h1 = l0 - a0, l1 - a1
                acc = log_u.numpy() < h1 - h0
                r0, r1 = h0, h1
            else:
                acc = log_u.numpy() < (l1 - l0) + (a1 - a0)
                r0, r1 = l0, l1
        if mask is not None:
            mask.numpy()[...] = acc
        if ret is not None:
            ret.numpy()[...] = np.where(acc, r1, r0)
        lp_cur.numpy()[acc] = l1[acc]
        if count is not None:
            count += int(acc.sum())

    def select_columns(self, mask, dst0, src0, dst1=None, src1=None, copy0=None):
        self._count("select_columns")
        m = mask.numpy().astype(bool)
        dst0.numpy()[:, m] = src0.numpy()[:, m]
        if dst1 is not None:
            dst1.numpy()[:, m] = src1.numpy()[:, m]
        if copy0 is not None:
            copy0.numpy()[...] = dst0.numpy()

    def blend_columns(self, mask, a, b, out):
        self._count("blend_columns")
        m = mask.numpy().astype(bool)
        out.numpy()[...] = np.where(m[None, :], b.numpy(), a.numpy())

    # -- MALA ------------------------------------------------------------------------------------
    def mala_propose(self, kind, state, theta, grad, theta_prop, eps, sqrt2eps):
        self._count("mala_propose")
        D, C = theta.shape
        for c in range(C):
            g = self._gen(kind, state, c)
            z = g.standard_normal(D)
            theta_prop.numpy()[:, c] = (theta.numpy()[:, c] + eps * grad.numpy()[:, c]) + sqrt2eps * z
            self._put(kind, state, c, g)

    def normals_chain_major(self, kind, state, zt, D, snapshot=None, max_workgroups=0):
        self._count("normals_chain_major")
        if snapshot is not None:
            snapshot.numpy()[...] = state.numpy()
        for c in range(zt.shape[0]):
            g = self._gen(kind, state, c)
            zt.numpy()[c, :D] = g.standard_normal(D)
            self._put(kind, state, c, g)

    def mala_propose_from_normals(self, theta, grad, z, theta_prop, eps, sqrt2eps):
        theta_prop.numpy()[...] = (theta.numpy() + eps * grad.numpy()) + sqrt2eps * z.numpy()

    def mala_logq(self, theta, grad, theta_prop, grad_prop, eps, lp_forward, lp_reverse):
        self._count("mala_logq")
        th, g, thp, gp = theta.numpy(), grad.numpy(), theta_prop.numpy(), grad_prop.numpy()
        xf = (thp - th) - eps * g
        xr = (th - thp) - eps * gp
        sf = np.zeros(th.shape[1])
        sr = np.zeros(th.shape[1])
        for d in range(th.shape[0]):
            sf = sf + xf[d] * xf[d]
            sr = sr + xr[d] * xr[d]
        k = -0.25 / eps
        lp_forward.numpy()[...] = k * sf
        lp_reverse.numpy()[...] = k * sr

    def mala_step_supported(self, C, D, ld):
        return C % 2 == 0 and ld % 2 == 0 and 0 < D <= 1024

    def mala_step_gaussian(self, lam, theta, theta_out, theta_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret, count):
        self._count("mala_step_gaussian")
        import torch

        kind = "iso_gaussian" if lam is None else "diag_gaussian"
        g, gp = torch.zeros_like(theta), torch.zeros_like(theta_prop)
        self.target_grad(kind, lam, theta, g, None)
        self.target_grad(kind, lam, theta_prop, gp, None)
        self.mala_step(theta, theta_out, g, theta_prop, gp, lp, lp_prop, log_u, zt_next, eps, sqrt2eps, mask, ret, count)

    def mala_step(self, theta, theta_out, grad, theta_prop, grad_prop, lp, lp_prop, log_u, zt_next, eps, sqrt2eps,
                  mask, ret, count):
        self._count("mala_step")
        th, g, thp, gp = theta.numpy(), grad.numpy(), theta_prop.numpy(), grad_prop.numpy()
        xf = (thp - th) - eps * g
        xr = (th - thp) - eps * gp
        sf = np.zeros(th.shape[1])
        sr = np.zeros(th.shape[1])
        for d in range(th.shape[0]):
            sf = sf + xf[d] * xf[d]
            sr = sr + xr[d] * xr[d]
        k = -0.25 / eps
        l0, l1 = lp.numpy().copy(), lp_prop.numpy()
        with np.errstate(invalid="ignore"):
            acc = log_u.numpy() < (l1 - l0) + (k * sr - k * sf)
        new_th = np.where(acc[None, :], thp, th)
        new_g = np.where(acc[None, :], gp, g)
        theta_out.numpy()[...] = new_th
        grad.numpy()[...] = new_g
        lp.numpy()[...] = np.where(acc, l1, l0)
        if ret is not None:
            ret.numpy()[...] = lp.numpy()
        if mask is not None:
            mask.numpy()[...] = acc
        if count is not None:
            count += int(acc.sum())
        if zt_next is not None:
            D = th.shape[0]
            theta_prop.numpy()[...] = (new_th + eps * new_g) + sqrt2eps * zt_next.numpy()[:, :D].T

    # -- targets ----------------------------------------------------------------------------------
    def target_grad(self, kind, params, theta, grad, logp, n_dev=None):
        self._count("target_grad")
        if n_dev is not None:
            n = self._lanes(theta.shape[1], n_dev)
            theta = theta[:, :n]
            grad = None if grad is None else grad[:, :n]
            logp = None if logp is None else logp[:n]
        th = theta.numpy()
        D, C = th.shape
        if kind in ("iso_gaussian", "diag_gaussian"):
            lt = th if kind == "iso_gaussian" else params.numpy()[:, None] * th
            if grad is not None:
                grad.numpy()[...] = -lt
            if logp is not None:
                s = np.zeros(C)
                for d in range(D):
                    s = s + th[d] * lt[d]
                logp.numpy()[...] = -0.5 * s
        elif kind == "funnel":
            v = th[0]
            s = np.zeros(C)
            for d in range(1, D):
                s = s + th[d] * th[d]
            ev = np.exp(-v)
            hn = 0.5 * (D - 1)
            he = 0.5 * ev
            if logp is not None:
                logp.numpy()[...] = ((-(v * v) / 18.0) - hn * v) - he * s
            if grad is not None:
                grad.numpy()[0] = ((-v / 9.0) - hn) + he * s
                grad.numpy()[1:] = -(ev[None, :] * th[1:])
        else:
            raise KeyError(kind)

    def hmc_trajectory_funnel(self, theta_in, rho, grad_in, theta_out, grad_out, logp_out, kin_out, metric, eps, steps):
        """bk_hmc_trajectory_funnel: hmc.py:40-53 on the funnel, energies of the end point; rho <- minus the end momentum."""
        self._count("hmc_trajectory_funnel")
        import torch

        half = 0.5 * eps
        th = theta_in.clone()
        t = self._mt(metric, grad_in.numpy())
        r = rho.numpy().copy()
        r = r + (-half) * t
        r = r + eps * t
        th.numpy()[...] = th.numpy() + eps * r
        g = torch.zeros_like(th)
        for _ in range(int(steps) - 1):
            self.target_grad("funnel", None, th, g, None)
            t = self._mt(metric, g.numpy())
            r = r + eps * t
            th.numpy()[...] = th.numpy() + eps * r
        self.target_grad("funnel", None, th, g, logp_out)
        t = self._mt(metric, g.numpy())
        r = -(r + half * t)
        mr = r if metric is None else metric.numpy()[:, None] * r
        kin = np.zeros(r.shape[1])
        for d in range(r.shape[0]):
            kin = kin + r[d] * mr[d]
        kin_out.numpy()[...] = 0.5 * kin
        rho.numpy()[...] = r
        theta_out.numpy()[...] = th.numpy()
        grad_out.numpy()[...] = g.numpy()

    def leapfrog_step_funnel(self, theta, rho, metric, h, n_dev=None):
        """bk_leapfrog_step_funnel: {gradient, kick, drift} of one leapfrog step, in place."""
        self._count("leapfrog_step_funnel")
        import torch

        g = torch.zeros_like(theta)
        self.target_grad("funnel", None, theta, g, None, n_dev=n_dev)
        self.kick_drift(theta, theta, rho, rho, g, metric, h, False, 0.0, True, h, n_dev=n_dev)

    def hmc_trajectory_gaussian(self, theta_in, theta_out, rho_in, rho_out, lam, metric, eps, steps):
        th, r = theta_in.numpy().copy(), rho_in.numpy().copy()
        lamv = None if lam is None else lam.numpy()[:, None]
        grad = lambda x: -x if lamv is None else -(lamv * x)
        half = 0.5 * eps
        t = self._mt(metric, grad(th))
        r = r + (-half) * t
        for _ in range(steps):
            r = r + eps * t
            th = th + eps * r
            t = self._mt(metric, grad(th))
        r = r + half * t
        theta_out.numpy()[...] = th
        rho_out.numpy()[...] = r

    @staticmethod
    def _quarter_sum(x):
        """sum over d of x[d, c], sequential in d like this file's other per-chain sums (the HIP
        kernels sum four contiguous quarters and combine them: the same value to ~1e-16 relative)."""
        s = np.zeros(x.shape[1])
        for d in range(x.shape[0]):
            s = s + x[d]
        return s

    def hmc_draw_gaussian(self, theta_in, theta_out, rho_in, zt, lam, metric, eps, steps, part, kin0, kin1, lp_out,
                          accept=None):
        self._count("hmc_draw_gaussian")
        D = theta_in.shape[0]
        r0 = torch.from_numpy(0.0 + 1.0 * zt.numpy()[:, :D].T.copy()) if zt is not None else rho_in
        r1 = torch.empty_like(theta_in)
        m = (lambda v: v) if metric is None else (lambda v: metric.numpy()[:, None] * v)
        if kin0 is not None:
            kin0.numpy()[...] = 0.5 * self._quarter_sum(r0.numpy() * m(r0.numpy()))
        self.hmc_trajectory_gaussian(theta_in, theta_out, r0, r1, lam, metric, eps, steps)
        kin1.numpy()[...] = 0.5 * self._quarter_sum(r1.numpy() * m(r1.numpy()))
        th = theta_out.numpy()
        lt = th if lam is None else lam.numpy()[:, None] * th
        lp_out.numpy()[...] = -0.5 * self._quarter_sum(th * lt)
        if accept is not None:
            lp_cur, log_u, mask, ret, count = accept
            if kin0 is None:
                kin0 = torch.from_numpy(0.5 * self._quarter_sum(r0.numpy() * m(r0.numpy())))
            self.mh_accept(0, lp_cur, kin0, lp_out, kin1, log_u, mask, ret, count)

    @staticmethod
    def _lanes(n, n_dev):
        return n if n_dev is None else min(int(n), int(n_dev[0]))

    _prop_target = ("funnel", None)  # (which target the shared proposal body evaluates)

    def dr_proposal_gaussian(self, lam, *args, **kw):
        """bk_dr_proposal_gaussian_job: the funnel proposal's body on a separable Gaussian."""
        keep, self._prop_target = self._prop_target, (("iso_gaussian", None) if lam is None else ("diag_gaussian", lam))
        try:
            self.dr_proposal_funnel(*args, **kw)
        finally:
            self._prop_target = keep

    def leapfrog_step_gaussian(self, lam, theta, rho, metric, h, n_dev=None):
        self._count("leapfrog_step_gaussian")
        import torch

        g = torch.zeros_like(theta)
        self.target_grad("iso_gaussian" if lam is None else "diag_gaussian", lam, theta, g, None, n_dev=n_dev)
        self.kick_drift(theta, theta, rho, rho, g, metric, h, False, 0.0, True, h, n_dev=n_dev)

    def dr_proposal_funnel(self, theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out, logp_out,
                           kin_out, metric, h, steps, n_dev=None, lanes_out=None, lanes_total=None, level=None, job=None,
                           ghost=None, ghost0=None):
        if grad_in is None:   # no cached gradient of the source point: the launch evaluates it
            grad_in = torch.zeros_like(theta_in)
            kind_, params_ = self._prop_target
            self.target_grad(kind_, params_, theta_in, grad_in, None)
        if grad_out is None:  # ... and does not store the end point's
            grad_out = torch.empty_like(theta_out)
        # a scatter job runs BESIDE the trajectories on the device (disjoint memory): do it afterwards here, so
        # that a job which overlapped the proposal's inputs or outputs would be noticed
        if job is not None or ghost is not None or ghost0 is not None:
            try:
                self.dr_proposal_funnel(theta_in, rho_in, grad_in, src_index, theta_out, rho_out, grad_out,
                                        logp_out, kin_out, metric, h, steps, n_dev, lanes_out, lanes_total, level)
                if ghost0 is not None:
                    # the produced level's first ghost as a launch of its own into scratch arrays, then
                    # dr_accept_prob_ghost[_next] against the produced level
                    assert level is not None and (ghost0["next_index"] is None or ghost is None)
                    g0 = ghost0
                    n_l = theta_out.shape[1]
                    th, rh, gr = (torch.empty_like(t) for t in (theta_out, rho_out, grad_out))
                    lp, kn, gH, gh, ga = (torch.empty(n_l, dtype=torch.float64) for _ in range(5))
                    glive = torch.empty(n_l, dtype=torch.uint8)
                    self.dr_proposal_funnel(theta_out, rho_out, grad_out, None, th, rh, gr, lp, kn, metric, g0["h"],
                                            g0["steps"], n_dev, g0["lanes_out"], g0["lanes_total"], (gH, gh, glive))
                    args = (gH, level[0], gh, level[1], None, g0["prob_retry"], glive, ga, n_l, level[2], g0["parent_a"])
                    if g0["next_index"] is None:
                        self.dr_accept_prob_ghost(*args, n_dev=n_dev)
                    else:
                        self.dr_accept_prob_ghost_next(*args, g0["next_index"], g0["next_count"], n_dev=n_dev)
                if ghost is not None:
                    gl = ghost
                    n_l = theta_out.shape[1]
                    args = (level[0], gl["parent_H"], level[1], gl["parent_h"], src_index, gl["prob_retry"], level[2],
                            gl["a_out"], n_l, gl["parent_live"], gl["parent_a"])
                    if gl["next_index"] is None:
                        self.dr_accept_prob_ghost(*args, n_dev=n_dev)
                    else:
                        self.dr_accept_prob_ghost_next(*args, gl["next_index"], gl["next_count"], n_dev=n_dev)
                return
            finally:
                if job is not None:
                    self.scatter_columns(*job["args"], **job["kw"])
        n = self._lanes(theta_out.shape[1], n_dev)
        if lanes_out is not None:
            lanes_out[0] = n
        if lanes_total is not None:
            lanes_total[0] += n
        lvl_logp, lvl_kin = logp_out, kin_out
        if n < theta_out.shape[1]:  # device-side lane count: only the first n lanes of the outputs exist
            theta_out, rho_out, grad_out = theta_out[:, :n], rho_out[:, :n], grad_out[:, :n]
            logp_out, kin_out = logp_out[:n], kin_out[:n]
        if n == 0:
            return
        self.first_step_gather(theta_in, rho_in, grad_in, src_index, theta_out, rho_out, metric, h, 0.5 * h)
        kind, params = self._prop_target
        for _ in range(steps - 1):
            self.target_grad(kind, params, theta_out, grad_out, None)
            self.kick_drift(theta_out, theta_out, rho_out, rho_out, grad_out, metric, h, False, 0.0, True, h)
        self.target_grad(kind, params, theta_out, grad_out, logp_out)
        self.leapfrog_finish(rho_out, rho_out, grad_out, metric, 0.5 * h, True, kin_out)
        if level is not None:
            self.dr_level_begin(lvl_logp, lvl_kin, level[0], level[1], level[2], n)

    def dense_metric_apply(self, M, X, Y):
        Y.numpy()[...] = M.numpy() @ X.numpy()

    def gemm_chains_work(self, R, K, C):
        return None

    def gemm_chains(self, A, X, Y, work=None):
        Y.numpy()[...] = A.numpy() @ X.numpy()

    def gemm_chains_logistic(self, A, X, Y, y_rows):
        self.gemm_chains(A, X, Y)
        self.logistic_residual(Y, y_rows, None)

    def logistic_residual(self, Z, y, part, segments=None):
        z = Z.numpy()
        yv = y.numpy()[:, None]
        if part is not None:
            S = part.shape[0]
            N = z.shape[0]
            rows = -(-max(N, 1) // S)
            ll = yv * z - np.logaddexp(0.0, z)
            for s in range(S):
                part.numpy()[s] = ll[s * rows:(s + 1) * rows].sum(axis=0)
        from scipy.special import expit

        z[...] = yv - expit(z)

    def logistic_finish(self, G, theta, part, inv_prior_var, t, grad, logp, loglik):
        th = theta.numpy()
        ll = None if part is None else part.numpy().sum(axis=0)
        if grad is not None:
            grad.numpy()[...] = t * G.numpy() + (-(inv_prior_var * th))
        if loglik is not None:
            loglik.numpy()[...] = ll
        if logp is not None:
            logp.numpy()[...] = t * ll + (-0.5 * inv_prior_var * (th * th).sum(axis=0))

    def dot_columns(self, x, y, scale, out):
        out.numpy()[...] = scale * np.einsum("dc,dc->c", x.numpy(), y.numpy())

    def resample_indices(self, weights, u, cdf_work, idx_out):
        w = weights.numpy()
        cdf = np.cumsum(w / w.sum())   # RandomState.choice(p = w / w.sum()): cdf = p.cumsum(); cdf /= cdf[-1]
        cdf_work.numpy()[...] = cdf
        idx = np.searchsorted(cdf / cdf[-1], u.numpy(), side="right")
        idx_out.numpy()[...] = np.minimum(idx, len(cdf) - 1)

    def gather_columns(self, index, src, dst):
        dst.numpy()[...] = src.numpy()[:, index.numpy()[: dst.shape[1]]]

    def relayout(self, src, dst):
        self._count("relayout")
        dst.copy_(src)

    # -- diagnostics ----------------------------------------------------------------------------------
    def welford_update(self, mean, m2, theta, n):
        x, mu = theta.numpy(), mean.numpy()
        delta = x - mu
        mu2 = mu + delta / float(n)
        m2.numpy()[...] = m2.numpy() + delta * (x - mu2)
        mean.numpy()[...] = mu2

    def rhat_partials(self, mean, m2, n, center, out):
        mu = mean.numpy()
        D = mu.shape[0]
        o = out.numpy().reshape(-1)
        o[0:D] = mu.sum(axis=1)
        o[D:2 * D] = (m2.numpy() / float(n - 1)).sum(axis=1)
        if center is not None:
            o[2 * D:3 * D] = ((mu - center.numpy()[:, None]) ** 2).sum(axis=1)

    def chain_mean_var(self, x, lengths, mean, var):
        X = x.numpy()
        for c in range(X.shape[1]):
            n = X.shape[0] if lengths is None else int(lengths[c])
            col = X[:n, c]
            mean[c] = col.mean()
            if var is not None:
                var[c] = col.var(ddof=1)

    def rank_normalize(self, rank, S, out):
        import scipy.stats

        out.numpy().reshape(-1)[...] = scipy.stats.norm.ppf((rank.numpy().reshape(-1) - 0.325) / (S - 0.25))

    def sort_by_key(self, keys, vals):
        ko, order = torch.sort(keys, stable=True)
        return ko, vals[order]

    def count_below(self, sorted_keys, queries):
        return torch.searchsorted(sorted_keys.contiguous(), queries.contiguous(), right=False)

    def scatter_ranks(self, payload, base, out):
        out[payload] = base + torch.arange(1, payload.numel() + 1, dtype=out.dtype)

    def autocorr(self, x, out):
        from oracle import diagnostics as od

        for c in range(x.shape[1]):
            out.numpy()[:, c] = od.autocorr(x.numpy()[:, c])

    def autocorr_fft(self, x, out):
        self.autocorr(x, out)

    def end_pos_pairs(self, acor, out):
        a = acor.numpy()
        for c in range(a.shape[1]):
            n = 0
            while n + 1 < a.shape[0] and not (a[n, c] + a[n + 1, c] < 0):
                n += 2
            out[c] = n

    def iat_from_acor(self, acor, estimator, ess_out, iat_out=None):
        a = acor.numpy()
        N = a.shape[0]
        for c in range(a.shape[1]):
            total, prev_min, first, n = 0.0, 0.0, True, 0
            while n + 1 < N:
                pk = a[n, c] + a[n + 1, c]
                if first:
                    prev_min, first = pk, False
                    if estimator == 0:
                        total = pk
                    if pk < 0:
                        break
                    if estimator == 1:
                        total = pk
                else:
                    if pk < 0:
                        break
                    if estimator == 0:
                        prev_min = min(prev_min, pk)
                        total += prev_min
                    else:
                        total += pk
                n += 2
            it = 2.0 * total - 1.0
            if iat_out is not None:
                iat_out[c] = it
            ess_out[c] = N / it

    def ess(self, x, estimator, ess_out, iat_out=None):
        from oracle import diagnostics as od

        X = x.numpy()
        for c in range(X.shape[1]):
            it = od.iat_imse(X[:, c]) if estimator == 0 else od.iat_ipse(X[:, c])
            if iat_out is not None:
                iat_out[c] = it
            ess_out[c] = X.shape[0] / it

    # -- delayed rejection --------------------------------------------------------------------------
    def compact_indices(self, mask, n, idx_out, count_out, n_dev=None):
        n = self._lanes(n, n_dev)
        nz = np.nonzero(mask.numpy()[:n])[0]
        idx_out.numpy()[: len(nz)] = nz
        count_out.numpy()[0] = len(nz)

    @staticmethod
    def _joint(logp, kin):
        return -((-logp) + kin)

    def dr_begin(self, logp, kin, cur_H, cur_h, rej, alive):
        cur_H.numpy()[...] = self._joint(logp.numpy(), kin.numpy())
        cur_h.zero_()
        rej.zero_()
        alive.fill_(1)

    def dr_retry_test(self, kind, state, rej, prob_retry, alive):
        for c in range(alive.shape[0]):
            if not alive[c]:
                continue
            g = self._gen(kind, state, c)
            with np.errstate(divide="ignore", invalid="ignore"):
                lu = np.log(g.uniform())
                retry = prob_retry * rej.numpy()[c]
                if not lu < retry:
                    alive[c] = 0
            self._put(kind, state, c, g)

    def dr_level_begin(self, logp, kin, H, h, live, n, n_dev=None):
        n = self._lanes(n, n_dev)
        H.numpy()[:n] = self._joint(logp.numpy()[:n], kin.numpy()[:n])
        h.numpy()[:n] = 0.0
        live.numpy()[:n] = 1

    def dr_ghost_update(self, ga, sub_index, m, h, live, a, n_dev=None):
        m = self._lanes(m, n_dev)
        for j in range(m):
            p = j if sub_index is None else int(sub_index[j])
            g = ga.numpy()[j]
            if g == 0:
                a[p] = -np.inf
                live[p] = 0
            else:
                h.numpy()[p] = h.numpy()[p] + np.log1p(-np.exp(g))

    def dr_accept_prob(self, H, cur_H, h, cur_h, cur_index, prob_retry, live, a, n, n_dev=None):
        n = self._lanes(n, n_dev)
        for j in range(n):
            if not live[j]:
                continue
            p = j if cur_index is None else int(cur_index[j])
            ph, ch = h.numpy()[j], cur_h.numpy()[p]
            with np.errstate(invalid="ignore"):
                frac = ((H.numpy()[j] - cur_H.numpy()[p]) + (ph - ch)) + (prob_retry * ph - prob_retry * ch)
            a[j] = frac if frac < 0 else 0.0

    def dr_accept_test(self, kind, state, chain_index, a, H, n, cur_H, cur_h, rej, alive, accepted, n_dev=None):
        n = self._lanes(n, n_dev)
        for j in range(n):
            c = j if chain_index is None else int(chain_index[j])
            g = self._gen(kind, state, c)
            with np.errstate(divide="ignore"):
                lu = np.log(g.uniform())
            self._put(kind, state, c, g)
            aj = a.numpy()[j]
            if lu < aj:
                accepted[j] = 1
                cur_H[c] = H[j]
                alive[c] = 0
            else:
                accepted[j] = 0
                with np.errstate(divide="ignore"):
                    r = np.log1p(-np.exp(aj))
                rej.numpy()[c] = r
                cur_h.numpy()[c] = cur_h.numpy()[c] + r

    def dr_accept_prob_test(self, kind, state, chain_index, H, h, live, a, prob_retry, n, cur_H, cur_h, rej, alive,
                            accepted, n_dev=None):
        # (every chain is in the lane set once: evaluating all probabilities first changes nothing)
        self.dr_accept_prob(H, cur_H, h, cur_h, chain_index, prob_retry, live, a, n, n_dev)
        self.dr_accept_test(kind, state, chain_index, a, H, n, cur_H, cur_h, rej, alive, accepted, n_dev)

    def dr_accept_prob_ghost(self, H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, parent_live, parent_a,
                             n_dev=None):
        # (every parent lane has one ghost lane)
        self.dr_accept_prob(H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, n_dev)
        self.dr_ghost_update(a, sub_index, n, parent_h, parent_live, parent_a, n_dev)

    # The appending forms build their lists in REVERSE lane order here: on the device the order depends on
    # wavefront timing, and nothing may depend on it.
    def dr_begin_retry(self, kind, state, logp, kin, cur_H, cur_h, rej, alive, prob_retry, counters, draw_counter=None):
        self.dr_begin(logp, kin, cur_H, cur_h, rej, alive)
        self.dr_retry_test(kind, state, rej, prob_retry, alive)
        if counters is not None:
            counters.zero_()
        if draw_counter is not None:
            draw_counter += 1

    def diag_job(self, theta, n_dev, welford=None, record=None):
        return (theta, n_dev, welford, record)

    def dr_refresh_begin(self, kind, state, loc_in, loc_mul, scale, out, metric, kin_out, work, logp, cur_H, cur_h, rej,
                         alive, prob_retry, counters, draw_counter=None, side=None):
        if side is not None:  # (beside the generator on the device: it reads the current point and the count as the draw finds them)
            theta, n_dev, welford, record = side
            if welford is not None:
                self.welford_update_dev(welford[0], welford[1], theta, n_dev, welford[2])
            if record is not None:
                self.record_series_dev(theta, record[0], record[1], record[2], n_dev, record[3])
        self.momentum_refresh(kind, state, loc_in, loc_mul, scale, out, metric, kin_out, None, work)
        self.dr_begin_retry(kind, state, logp, kin_out, cur_H, cur_h, rej, alive, prob_retry, counters, draw_counter)

    def record_series_dev(self, theta, dims, logp, series, row_dev, row_offset):
        row = int(row_dev[0]) - int(row_offset)
        if 0 <= row < series.shape[1]:
            self.record_series(theta, dims, logp, series, row)

    def welford_update_dev(self, mean, m2, theta, n_dev, n_offset):
        self.welford_update(mean, m2, theta, int(n_dev[0]) - int(n_offset))

    def dr_accept_prob_test_next(self, kind, state, chain_index, H, h, live, a, prob_retry, n, cur_H, cur_h, rej, alive,
                                 accepted, next_index, next_count, n_dev=None):
        assert int(next_count[0]) == 0
        self.dr_accept_prob_test(kind, state, chain_index, H, h, live, a, prob_retry, n, cur_H, cur_h, rej, alive,
                                 accepted, n_dev)
        again = []
        for j in range(self._lanes(n, n_dev)):
            if accepted[j]:
                continue
            c = j if chain_index is None else int(chain_index[j])
            g = self._gen(kind, state, c)
            with np.errstate(divide="ignore", invalid="ignore"):
                lu = np.log(g.uniform())
                if lu < prob_retry * rej.numpy()[c]:
                    again.append(c)
                else:
                    alive[c] = 0
            self._put(kind, state, c, g)
        next_index.numpy()[: len(again)] = again[::-1]
        next_count.numpy()[0] = len(again)

    def dr_accept_prob_ghost_next(self, H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, parent_live,
                                  parent_a, next_index, next_count, n_dev=None):
        assert int(next_count[0]) == 0
        self.dr_accept_prob_ghost(H, parent_H, h, parent_h, sub_index, prob_retry, live, a, n, parent_live, parent_a, n_dev)
        m = self._lanes(n, n_dev)
        keep = [j if sub_index is None else int(sub_index[j]) for j in range(m)]
        keep = [p for p in keep if parent_live[p]]
        next_index.numpy()[: len(keep)] = keep[::-1]
        next_count.numpy()[0] = len(keep)

    def record_series(self, theta, dims, logp, series, row):
        for k, d in enumerate([] if dims is None else dims.tolist()):
            series[k, row].copy_(theta[d])
        if logp is not None:
            series[-1, row].copy_(logp)

    def scatter_job(self, *args, **kw):
        return {"args": args, "kw": kw}

    def ghost0(self, h, steps, parent_a, prob_retry, next_index=None, next_count=None, lanes_out=None, lanes_total=None):
        return dict(h=float(h), steps=int(steps), parent_a=parent_a, prob_retry=prob_retry, next_index=next_index,
                    next_count=next_count, lanes_out=lanes_out, lanes_total=lanes_total)

    def ghost_link(self, parent_H, parent_h, parent_live, parent_a, a_out, prob_retry, next_index=None, next_count=None):
        return dict(parent_H=parent_H, parent_h=parent_h, parent_live=parent_live, parent_a=parent_a, a_out=a_out,
                    prob_retry=prob_retry, next_index=next_index, next_count=next_count)

    def scatter_columns(self, mask, index, n, dsts, srcs, sdst=None, ssrc=None, n_dev=None):
        n = self._lanes(n, n_dev)
        for j in range(n):
            if not mask[j]:
                continue
            g = j if index is None else int(index[j])
            for d, s in zip(dsts, srcs):
                d.numpy()[:, g] = s.numpy()[:, j]
            if sdst is not None:
                sdst[g] = ssrc[j]
