"""Likelihood gradients for the LMC model (mirror of reference
runlmc/lmc/likelihood.py:20-134), batched on the device.

The reference builds one dK operator per hyper-parameter and spends N+1
operator products on each (likelihood.py:48-96,112-131).  Every one of those
dK is  W (dB (x) T) W^T  for a D x D matrix dB and a Toeplitz T that is either
k_q or dk_q/dtheta, so with u~ = W^T u reshaped D x m

    u^T dK v = sum_ab dB[a, b] * P_T(u, v)[a, b],   P_T(u, v)[a, b] = u~_a . T v~_b.

One D x D matrix  G_T = (P_T(alpha, alpha) - mean_i P_T(K^-1 r_i, r_i)) / 2
per top row T therefore yields every gradient that involves T:

    d/dA_q     = A_q (G_q + G_q^T)          (dB = e_j a_i^T + a_i e_j^T)
    d/dkappa_q = diag(G_q)                  (dB = e_i e_i^T)
    d/dtheta   = sum_ab B_q[a, b] G'[a, b]  (T = dk_q/dtheta, dB = B_q)
    d/deps_d   = (sum_{i in d} alpha_i^2 - mean_probe sum_{i in d} s_i r_i) / 2.

Cost: (Q + sum_q p_q) batched Toeplitz products over the N+1 vectors instead
of P (N+1) full operator products.  The probe sums are what the single
all-reduce of a multi-GPU step carries.
"""
import numpy as np
import torch

from .._native import GridOp, cross_dots, segment_dots
from .._lib import as_f64
from ..util.dist import all_reduce_sum_


class LMCLikelihood:
    """Interface the functional kernel pulls gradients from."""

    def __init__(self, functional_kernel, Ys):
        self.functional_kernel = functional_kernel
        self.y = np.hstack(Ys)
        self.lens = [len(Y) for Y in Ys]

    def alpha(self):
        raise NotImplementedError

    def coreg_vec_gradients(self):
        raise NotImplementedError

    def coreg_diags_gradients(self):
        raise NotImplementedError

    def kernel_gradients(self):
        raise NotImplementedError

    def noise_gradient(self):
        raise NotImplementedError


_GRAD_OPS = {}


def _grad_operator(D, m, ntops, device_index, sizes=None):
    """Top-row-only operators are parameter free apart from their spectra, so
    one handle per grid shape is reused across optimiser steps."""
    key = (D, m, device_index, sizes)
    op = _GRAD_OPS.get(key)
    if op is None or op.max_tops < ntops:
        op = GridOp(D, m, ntops, device_index=device_index, sizes=sizes)
        _GRAD_OPS[key] = op
    return op


class ApproxLMCLikelihood(LMCLikelihood):
    def __init__(self, functional_kernel, grid_kern, grid_dists,
                 interpolants, Ys, deriv, probes=None):
        super().__init__(functional_kernel, Ys)
        fk = functional_kernel
        self.K = grid_kern
        self.interpolants = interpolants
        self.materialized_kernels = fk.eval_kernels(grid_dists)
        self.materialized_grads = fk.eval_kernel_gradients(grid_dists)
        if probes is None:
            self.deriv = deriv.generate(self.K, self.y)
        else:
            self.deriv = deriv.generate(self.K, self.y, rs=probes)
        self._parts = None

    def alpha(self):
        return self.deriv.alpha

    # -- the likelihood value itself (reference models/interpolated_llgp.py:
    #    262-290 computes these on the model, with a DENSE Cholesky log-det of
    #    the exact kernel; here the log-det is matrix-free) ----------------------
    def normal_quadratic(self):
        """y^T K^-1 y with the Krylov alpha (interpolated_llgp.py:278-285)."""
        return float(self.y.dot(self.deriv.alpha))

    def log_det_K(self):
        """Stochastic Lanczos quadrature estimate of log det K~ from the probe
        solves' own Lanczos coefficients (no extra MVMs)."""
        return self.deriv.logdet_K()

    def log_likelihood(self):
        """-(log det K + y^T alpha + n log 2 pi) / 2
        (interpolated_llgp.py:287-290)."""
        n = len(self.y)
        return -0.5 * (self.log_det_K() + self.normal_quadratic() +
                       n * np.log(2 * np.pi))

    # -- Gram terms in the coefficient space of a polynomial-form operator ----------
    # (False: always the streaming products below)
    COEFFICIENT_GRAMS = True

    def _coefficient_grams(self, skiop, term, grid, qs, U, V):
        """P[t, v, a, b] = u~_a . T_t v~_b for the term's top rows k_q and dk_q/dtheta WITHOUT a
        grid vector, when the operator is wholly in the polynomial form (round 6): every such
        T is Phi C_T Phi^T to the accepted 2e-13, so with c_u = Phi^T W^T u (D x r per vector:
        rl_ski_project, one pass over the batch)  u~_a . T v~_b = c_u[a]^T C_T c_v[b]  -- two
        projections and a tiny contraction instead of two interpolation products, one batched
        Toeplitz product and one Gram pass per top row.  The C of k_q come from the operator's
        own handle; the derivative rows are verified in a handle of their own (same grid, same
        basis: a smaller rank is the leading block).  None when anything is outside the form:
        the caller then runs the streaming products."""
        if term != 0 or len(skiop.grids) != 1 or grid.sizes is not None and len(grid.sizes) > 1:
            return None
        try:
            if not skiop.factor()[0] or skiop.factor_mode != 1:
                return None
        except NotImplementedError:
            return None
        R = grid.form()[0]
        if R <= 0 or grid.Q < len(qs):
            return None
        Cs = []
        for i in range(len(qs)):
            r, C = grid.poly_coeffs(i)
            if r != R:
                return None
            Cs.append(C)
        dt = [as_f64(np.ravel(g)) for q in qs for g in self.materialized_grads[q]]
        Rall = R
        if dt:
            D = grid.D
            gop = _grad_operator(D, grid.m, len(dt), grid.device_index, sizes=grid.sizes)
            gop.set_rank_hint(R)       # (derivative rows are no smoother than the rows themselves)
            gop.set_lmc(np.stack(dt), [None] * len(dt), [np.zeros(D)] * len(dt))
            for t in range(len(dt)):
                r, C = gop.poly_coeffs(t)
                if r == 0:
                    return None
                Rall = max(Rall, r)
                Cs.append(C)
        Cs = [np.pad(C, ((0, Rall - C.shape[0]),) * 2) for C in Cs]
        if Rall == R:
            cU, cV = skiop.project(U), skiop.project(V)
        else:
            # the derivative rows need a larger basis than the operator's own: interpolate to
            # the grid and project there on the first Rall polynomials (the basis is nested)
            cU = grid.project(skiop.apply_wt(U, term), Rall)
            cV = grid.project(skiop.apply_wt(V, term), Rall)
        Ct = torch.from_numpy(np.stack(Cs)).to(skiop.device)
        return torch.einsum('vai,tij,vbj->tvab', cU, Ct, cV).contiguous()

    # -- the batched partial sums ------------------------------------------------
    def _partials(self):
        if self._parts is not None:
            return self._parts
        fk = self.functional_kernel
        D, Q = fk.D, fk.Q
        dv = self.deriv
        skiop = self.K.device_operator()
        lib, dev = skiop.lib, skiop.device
        term_of = getattr(self.K, 'term_of', None) or {ad: 0 for ad in fk.active_dims}
        nloc = dv.rs_dev.shape[0]
        # alpha rides in a transform pair of its own (last of an odd batch, or
        # next to a zero row), as in the solves
        arow = dv.alpha_dev[None, :]
        if nloc % 2 == 0:
            U = torch.cat([dv.inv_rs_dev, arow], dim=0).contiguous()
            V = torch.cat([dv.rs_dev, arow], dim=0).contiguous()
            ia, probe_rows = nloc, slice(0, nloc)
        else:
            zero = torch.zeros_like(arow)
            U = torch.cat([arow, zero, dv.inv_rs_dev], dim=0).contiguous()
            V = torch.cat([arow, zero, dv.rs_dev], dim=0).contiguous()
            ia, probe_rows = 0, slice(2, 2 + nloc)
        nrow = U.shape[0]

        # per active-dimension set (= per grid = per term of the operator): the
        # top rows k_q and dk_q/dtheta of its kernels, one batched Toeplitz
        # product and one D x D Gram per top row
        owner = []            # (q, None) for k_q, (q, p) for dk_q/dtheta_p
        blocks = []           # P tensors (ntops_t, nrow, D, D)
        for ad, qs in fk.active_dims.items():
            term = term_of[ad]
            grid = skiop.grids[term]
            tops_t, own_t = [], []
            for q in qs:
                tops_t.append(as_f64(np.ravel(self.materialized_kernels[q])))
                own_t.append((q, None))
            for q in qs:
                for p_, g in enumerate(self.materialized_grads[q]):
                    tops_t.append(as_f64(np.ravel(g)))
                    own_t.append((q, p_))
            nt = len(tops_t)
            Pc = self._coefficient_grams(skiop, term, grid, qs, U, V) if self.COEFFICIENT_GRAMS else None
            if Pc is not None:
                owner += own_t
                blocks.append(Pc)
                continue
            gop = _grad_operator(D, grid.m, nt, grid.device_index, sizes=grid.sizes)
            gop.set_rank_hint(0)
            gop.set_lmc(np.stack(tops_t), [None] * nt, [np.zeros(D)] * nt)
            Ut = skiop.apply_wt(U, term)
            Vt = skiop.apply_wt(V, term)
            P = torch.empty((nt, nrow, D, D), dtype=torch.float64, device=dev)
            TV = torch.empty_like(Vt)
            for t in range(nt):
                gop.mvm(Vt, out=TV, top=t)
                P[t] = cross_dots(lib, Ut, TV, D, grid.m)
            owner += own_t
            blocks.append(P)
        P = torch.cat(blocks, dim=0)
        ntops = P.shape[0]
        offsets = torch.from_numpy(
            np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int32)).to(dev)
        seg = segment_dots(lib, U, V, offsets, D)          # (nrow, D)

        # ONE all-reduce: the probe sums of every rank and, from rank 0 only,
        # the alpha terms (alpha is the same vector everywhere, but the batch
        # it rides in differs from rank to rank and with it the kernels and
        # summation orders; taking rank 0's terms makes the assembled
        # gradient the same bits on every rank)
        from ..util.dist import rank_world
        rank0 = rank_world(dv._group)[0] == 0
        aP, aseg = P[:, ia].reshape(-1), seg[ia].reshape(-1)
        if not rank0:
            aP, aseg = torch.zeros_like(aP), torch.zeros_like(aseg)
        probe = torch.cat([P[:, probe_rows].sum(dim=1).reshape(-1),
                           seg[probe_rows].sum(dim=0).reshape(-1), aP, aseg])
        all_reduce_sum_(probe, dv._group)
        N = dv._n_it
        nP = ntops * D * D
        Psum = probe[:nP].reshape(ntops, D, D)
        ssum = probe[nP:nP + D]
        Pa = probe[nP + D:2 * nP + D].reshape(ntops, D, D)
        sa = probe[2 * nP + D:]
        Gall = (0.5 * (Pa - Psum / N)).cpu().numpy()
        noise = (0.5 * (sa - ssum / N)).cpu().numpy()
        G = [None] * Q
        Gd = [[None] * len(self.materialized_grads[q]) for q in range(Q)]
        for t, (q, p_) in enumerate(owner):
            if p_ is None:
                G[q] = Gall[t]
            else:
                Gd[q][p_] = Gall[t]
        self._parts = dict(G=G, Gd=Gd, noise=noise)
        return self._parts

    def coreg_vec_gradients(self):
        G = self._partials()['G']
        return [a.dot(G[q] + G[q].T)
                for q, a in enumerate(self.functional_kernel.coreg_vecs)]

    def coreg_diags_gradients(self):
        G = self._partials()['G']
        return [np.diag(G[q]).copy() for q in range(self.functional_kernel.Q)]

    def kernel_gradients(self):
        Gd = self._partials()['Gd']
        return [[float(np.sum(B * Gqp)) for Gqp in Gd[q]]
                for q, B in enumerate(self.functional_kernel.coreg_mats())]

    def noise_gradient(self):
        return self._partials()['noise'].copy()
