"""Thin object wrappers over the C ABI handles (device-resident state).

These are the only places that call into the native library.  torch tensors
are used purely as device-memory handles (allocation + lifetime + streams).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import host_ptr, dev_ptr, as_f64


def _vec_batch(lib, X, width, device):
    """Accept (width,) / (k, width) numpy or torch input, return a contiguous
    float64 tensor of shape (k, width) on `device` plus a flag telling
    whether the input was a single vector."""
    if isinstance(X, torch.Tensor):
        t = X
        if t.dtype != torch.float64:
            raise TypeError('device vectors must be float64')
    else:
        t = torch.from_numpy(as_f64(X))
    single = t.dim() == 1
    if single:
        t = t.unsqueeze(0)
    if t.dim() != 2 or t.shape[1] != width:
        raise ValueError('expected vectors of length %d, got shape %s'
                         % (width, tuple(X.shape)))
    return t.to(device).contiguous(), single


class GridOp:
    """Device handle of K_UU = sum_q B_q (x) T_q  (include/runlmc_hip.h)."""

    def __init__(self, D, m, max_tops, device_index=0, lib=None, sizes=None):
        """`m` grid points per output; `sizes=(m1, m2)` (m1*m2 == m) makes the
        kernel matrices BTTB on a two-dimensional grid."""
        self.lib = lib or _lib.get_library()
        self.D, self.m, self.max_tops = int(D), int(m), int(max_tops)
        self.device = self.lib.torch_device(device_index)
        self.device_index = device_index
        self._h = ctypes.c_void_p()
        self.sizes = None if sizes is None else tuple(int(v) for v in sizes)
        if self.sizes is None or len(self.sizes) == 1:
            self.lib.call('rl_gridop_create', device_index, self.D, self.m,
                          self.max_tops, ctypes.byref(self._h))
        elif len(self.sizes) == 2:
            if self.sizes[0] * self.sizes[1] != self.m:
                raise ValueError('sizes %s do not multiply to m = %d' % (self.sizes, self.m))
            self.lib.call('rl_gridop_create_2d', device_index, self.D,
                          self.sizes[0], self.sizes[1], self.max_tops,
                          ctypes.byref(self._h))
        else:
            raise NotImplementedError(
                'grids of more than two dimensions have no device path')
        info = [ctypes.c_int() for _ in range(5)]
        self.lib.call('rl_gridop_info', self._h, *[ctypes.byref(i) for i in info])
        self.L, self.N1, self.N2, self.colsA, self.rowsB = [i.value for i in info]
        self.Q = 0

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.cdll.rl_gridop_destroy(h)
            self._h = ctypes.c_void_p()

    @property
    def handle(self):
        return self._h

    @property
    def width(self):
        return self.D * self.m

    def form(self):
        """(rank, min_elements): rank r > 0 when the current parameters run
        batches of >= min_elements elements in the polynomial-subspace form
        (csrc/rl_lowrank.h), 0 when on the transform kernels only."""
        r, n = ctypes.c_int(), ctypes.c_longlong()
        self.lib.call('rl_gridop_form', self._h, ctypes.byref(r), ctypes.byref(n))
        return r.value, n.value

    def top_forms(self):
        """(forms, structured): per top row 0 = transform kernels, 1 =
        polynomial-subspace form, 2 = recursive filter (csrc/rl_filter.h);
        structured is True when operator products above the batch gate need
        no transform."""
        forms = (ctypes.c_int * max(self.Q, 1))()
        st = ctypes.c_int()
        self.lib.call('rl_gridop_top_forms', self._h, forms, ctypes.byref(st))
        return [forms[q] for q in range(self.Q)], bool(st.value)

    def form_stats(self, q):
        """What the set-time verification of the polynomial form measured for top row q:
        (trial ratio, tail ratio, estimate of ||T - Phi C Phi^T||_2, estimate of ||T||_2)
        -- include/runlmc_hip.h: rl_gridop_form_stats."""
        out = np.zeros(4)
        self.lib.call('rl_gridop_form_stats', self._h, int(q), host_ptr(out))
        return tuple(float(v) for v in out)

    def poly_coeffs(self, q):
        """(rank, C): C = Phi^T T_q Phi (rank x rank, symmetric) when top row q is in the
        polynomial form for the current parameters, else (0, None) -- include/runlmc_hip.h:
        rl_gridop_poly_coeffs (runs the pending verification)."""
        r = ctypes.c_int()
        buf = np.zeros(48 * 48)
        self.lib.call('rl_gridop_poly_coeffs', self._h, int(q), host_ptr(buf), buf.size,
                      ctypes.byref(r))
        if r.value == 0:
            return 0, None
        return r.value, buf[:r.value * r.value].reshape(r.value, r.value).copy()

    def project(self, G, rank):
        """(k, D, rank) tensor of Phi_rank^T g per output block of the GRID vectors G (k, D*m):
        rl_gridop_project."""
        k = G.shape[0]
        out = torch.empty((k, self.D, int(rank)), dtype=torch.float64, device=self.device)
        self.lib.call('rl_gridop_project', self._h, dev_ptr(G.contiguous()), k, int(rank),
                      dev_ptr(out), self.lib.stream_ptr(self.device))
        return out

    def set_rank_hint(self, rank):
        """First basis size the verification of the polynomial form tries (0: from 24 up)."""
        self.lib.call('rl_gridop_set_rank_hint', self._h, int(rank))

    def set_form_gate(self, min_elements):
        """Smallest batch (nvec*D*m elements) run in the polynomial form;
        0 = every batch, negative = the library default."""
        self.lib.call('rl_gridop_set_form_gate', self._h, int(min_elements))

    def _tops(self, tops):
        tops = as_f64(tops)
        if tops.ndim != 2 or tops.shape[1] != self.m:
            raise ValueError('tops must have shape (Q, %d), got %s'
                             % (self.m, tops.shape))
        return tops

    def set_lmc(self, tops, coreg_vecs, coreg_diags):
        """B_q = A_q^T A_q + diag(kappa_q); coreg_vecs[q] is (R_q, D) (or
        empty), coreg_diags[q] is (D,)."""
        tops = self._tops(tops)
        Q = tops.shape[0]
        if len(coreg_vecs) != Q or len(coreg_diags) != Q:
            raise ValueError('need one coreg_vec block and one coreg_diag per kernel')
        rows, ranks = [], []
        for a in coreg_vecs:
            a = np.zeros((0, self.D)) if a is None else np.atleast_2d(as_f64(a))
            if a.size and a.shape[1] != self.D:
                raise ValueError('coreg_vec block must be (R, %d)' % self.D)
            if a.size:
                # all-zero rows (independent-GP kernels carry one) add nothing
                a = a[np.any(a != 0.0, axis=1)]
            ranks.append(a.shape[0] if a.size else 0)
            if a.size:
                rows.append(a)
        vecs = (np.ascontiguousarray(np.vstack(rows)) if rows
                else np.zeros((0, self.D)))
        diags = np.ascontiguousarray(
            np.vstack([as_f64(k).reshape(1, -1) for k in coreg_diags]))
        if diags.shape != (Q, self.D):
            raise ValueError('coreg_diags must be Q x D')
        ranks = np.ascontiguousarray(np.array(ranks, dtype=np.int32))
        self.lib.call('rl_gridop_set_lmc', self._h, Q, host_ptr(tops),
                      host_ptr(ranks), host_ptr(vecs) if vecs.size else None,
                      host_ptr(diags))
        self.Q = Q

    def set_dense(self, tops, Bs):
        tops = self._tops(tops)
        Q = tops.shape[0]
        Bs = as_f64(Bs)
        if Bs.shape != (Q, self.D, self.D):
            raise ValueError('B must have shape (Q, D, D)')
        self.lib.call('rl_gridop_set_dense', self._h, Q, host_ptr(tops),
                      host_ptr(Bs))
        self.Q = Q

    def mvm(self, X, out=None, top=None):
        """Device-side product; X: (k, D*m) tensor on self.device."""
        k = X.shape[0]
        if out is None:
            out = torch.empty_like(X)
        sp = self.lib.stream_ptr(self.device)
        if top is None:
            self.lib.call('rl_gridop_mvm', self._h, dev_ptr(X), dev_ptr(out), k, sp)
        else:
            self.lib.call('rl_gridop_mvm_top', self._h, int(top), dev_ptr(X),
                          dev_ptr(out), k, sp)
        return out

    def matmat_host(self, X, top=None):
        """numpy in, numpy out; X is (D*m,) or (k, D*m) (vectors as ROWS)."""
        t, single = _vec_batch(self.lib, X, self.width, self.device)
        y = self.mvm(t, top=top).cpu().numpy()
        return y[0] if single else y

    def spectrum(self, q):
        out = np.empty(self.L)
        self.lib.call('rl_gridop_spectrum_host', self._h, int(q), host_ptr(out))
        return out


class SkiOp:
    """Device handle of K~ = W K_UU W^T + diag(eps)."""

    def __init__(self, gridop, W, WT):
        self.lib = gridop.lib
        self.grid = gridop
        self.device = gridop.device
        n, ng = W.shape
        if ng != gridop.width:
            raise ValueError('W has %d columns, grid operator has %d points'
                             % (ng, gridop.width))
        if WT.shape != (ng, n):
            raise ValueError('WT must be the transpose of W')
        W = W.tocsr()
        WT = WT.tocsr()
        self.n = int(n)
        arrs = [np.ascontiguousarray(W.indptr, dtype=np.int32),
                np.ascontiguousarray(W.indices, dtype=np.int32),
                as_f64(W.data),
                np.ascontiguousarray(WT.indptr, dtype=np.int32),
                np.ascontiguousarray(WT.indices, dtype=np.int32),
                as_f64(WT.data)]
        self._h = ctypes.c_void_p()
        self.lib.call('rl_ski_create', gridop.handle, self.n,
                      *[host_ptr(a) for a in arrs], ctypes.byref(self._h))
        self.grids = [gridop]          # grid operator of every term (kept alive)

    @staticmethod
    def _csr_arrays(W, WT):
        W, WT = W.tocsr(), WT.tocsr()
        return [np.ascontiguousarray(W.indptr, dtype=np.int32),
                np.ascontiguousarray(W.indices, dtype=np.int32),
                as_f64(W.data),
                np.ascontiguousarray(WT.indptr, dtype=np.int32),
                np.ascontiguousarray(WT.indices, dtype=np.int32),
                as_f64(WT.data)]

    def add_term(self, gridop, W, WT):
        """Add W_t K_t W_t^T for kernels on another active-dimension set;
        returns the term index."""
        if W.shape != (self.n, gridop.width) or WT.shape != (gridop.width, self.n):
            raise ValueError('interpolant shapes do not match the operator')
        arrs = self._csr_arrays(W, WT)
        self.lib.call('rl_ski_add_term', self._h, gridop.handle,
                      *[host_ptr(a) for a in arrs])
        self.grids.append(gridop)
        return len(self.grids) - 1

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.cdll.rl_ski_destroy(h)
            self._h = ctypes.c_void_p()

    @property
    def handle(self):
        return self._h

    def set_noise(self, noise, lens):
        noise = as_f64(noise)
        lens = np.ascontiguousarray(np.asarray(lens), dtype=np.int32)
        if noise.shape != (self.grid.D,) or lens.shape != (self.grid.D,):
            raise ValueError('noise and lens must have one entry per output')
        self.lib.call('rl_ski_set_noise', self._h, host_ptr(noise), host_ptr(lens))

    def factor(self):
        """(available, logdet, cond): builds / refreshes the Woodbury factorisation of
        K~ = F M F^T + diag(eps) for the current parameters (include/runlmc_hip.h:
        rl_ski_factor).  available is False when some top row is not in the polynomial
        form (or the operator is otherwise outside it); `reason` then says why."""
        av, ld, cond = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        self.lib.call('rl_ski_factor', self._h, ctypes.byref(av), ctypes.byref(ld),
                      ctypes.byref(cond))
        self.factor_reason = ('' if av.value else
                              self.lib.cdll.rl_last_error().decode())
        # 1: the factorisation is K~^-1 (every top row in the polynomial form); 2: it inverts the
        # operator's projection on the polynomial subspace -- a preconditioner (solve_pcg)
        self.factor_mode = int(av.value)
        return bool(av.value), float(ld.value), float(cond.value)

    def project(self, X):
        """(k, D, r) tensor of Phi^T W^T x per output on the orthonormal polynomials of the
        operator's polynomial form (rl_ski_project); NotImplementedError outside the form."""
        k = X.shape[0]
        r = ctypes.c_int()
        flat = torch.empty((k * self.grid.D * 48,), dtype=torch.float64, device=self.device)
        self.lib.call('rl_ski_project', self._h, dev_ptr(X.contiguous()), k, dev_ptr(flat),
                      ctypes.byref(r), self.lib.stream_ptr(self.device))
        return flat[:k * self.grid.D * r.value].reshape(k, self.grid.D, r.value)

    def precond_sample(self, W):
        """(P^1/2 W rows, log det P): rows of identity covariance (+-1 probes) become rows with the
        covariance P of the operator's current factorisation (rl_ski_precond_sample)."""
        W = W.contiguous()
        out = torch.empty_like(W)
        ld = ctypes.c_double()
        self.lib.call('rl_ski_precond_sample', self._h, dev_ptr(W), dev_ptr(out), W.shape[0],
                      ctypes.byref(ld), self.lib.stream_ptr(self.device))
        return out, float(ld.value)

    def mvm(self, X, out=None):
        if out is None:
            out = torch.empty_like(X)
        self.lib.call('rl_ski_mvm', self._h, dev_ptr(X), dev_ptr(out),
                      X.shape[0], self.lib.stream_ptr(self.device))
        return out

    def apply_wt(self, X, term=0):
        out = torch.empty((X.shape[0], self.grids[term].width), dtype=torch.float64,
                          device=self.device)
        self.lib.call('rl_ski_apply_wt_term', self._h, int(term), dev_ptr(X),
                      dev_ptr(out), X.shape[0], self.lib.stream_ptr(self.device))
        return out

    def apply_w(self, G, term=0):
        out = torch.empty((G.shape[0], self.n), dtype=torch.float64,
                          device=self.device)
        self.lib.call('rl_ski_apply_w_term', self._h, int(term), dev_ptr(G),
                      dev_ptr(out), G.shape[0], self.lib.stream_ptr(self.device))
        return out

    def matmat_host(self, X):
        t, single = _vec_batch(self.lib, X, self.n, self.device)
        y = self.mvm(t).cpu().numpy()
        return y[0] if single else y


MINRES, CG = 0, 1
MINRES_RULE = 2     # MINRES with SciPy's own stopping tests off (include/runlmc_hip.h)


def solve_batch(ski, B, method=MINRES, tol=1e-4, check_every=100, maxiter=0,
                lanczos_cap=0):
    """Device batched solve K~ X = B.  B: (k, n) tensor on ski.device.
    Returns (X tensor, iterations int[k], residuals float[k], istop int[k]);
    with lanczos_cap > 0 (MINRES only) a fifth item, the (k, cap, 2) array of
    Lanczos coefficients (alfa_j, beta_{j+1}) of every system."""
    k = B.shape[0]
    X = torch.empty_like(B)
    iters = np.zeros(k, dtype=np.int32)
    istop = np.zeros(k, dtype=np.int32)
    resid = np.zeros(k, dtype=np.float64)
    if lanczos_cap > 0:
        if method not in (MINRES, MINRES_RULE):
            raise ValueError('Lanczos coefficients come from MINRES only')
        lz = np.zeros((k, int(lanczos_cap), 2), dtype=np.float64)
        ski.lib.call('rl_solve_batch_lanczos', ski.handle, dev_ptr(B), dev_ptr(X),
                     k, int(method), float(tol), int(check_every), int(maxiter),
                     host_ptr(iters), host_ptr(resid), host_ptr(istop),
                     host_ptr(lz), int(lanczos_cap), ski.lib.stream_ptr(ski.device))
        return X, iters, resid, istop, lz
    ski.lib.call('rl_solve_batch', ski.handle, dev_ptr(B), dev_ptr(X), k,
                 int(method), float(tol), int(check_every), int(maxiter),
                 host_ptr(iters), host_ptr(resid), host_ptr(istop),
                 ski.lib.stream_ptr(ski.device))
    return X, iters, resid, istop


def solve_direct(ski, B, tol=1e-4, max_refine=4):
    """Device batched solve K~ X = B through the polynomial form's Woodbury
    factorisation + iterative refinement to the reference's residual rule
    (include/runlmc_hip.h: rl_solve_direct).  Returns (X, iterations, residuals,
    istop); NotImplementedError when the operator has no such form."""
    k = B.shape[0]
    X = torch.empty_like(B)
    iters = np.zeros(k, dtype=np.int32)
    istop = np.zeros(k, dtype=np.int32)
    resid = np.zeros(k, dtype=np.float64)
    if k == 0:
        return X, iters, resid, istop
    ski.lib.call('rl_solve_direct', ski.handle, dev_ptr(B), dev_ptr(X), k, float(tol),
                 int(max_refine), host_ptr(iters), host_ptr(resid), host_ptr(istop),
                 ski.lib.stream_ptr(ski.device))
    return X, iters, resid, istop


def solve_pcg(ski, B, tol=1e-4, maxiter=0):
    """Device batched solve K~ X = B by conjugate gradients preconditioned with the Woodbury
    inverse of the operator's projection on the polynomial subspace (include/runlmc_hip.h:
    rl_solve_pcg).  Returns (X, iterations, residuals, istop)."""
    k = B.shape[0]
    X = torch.empty_like(B)
    iters = np.zeros(k, dtype=np.int32)
    istop = np.zeros(k, dtype=np.int32)
    resid = np.zeros(k, dtype=np.float64)
    if k == 0:
        return X, iters, resid, istop
    ski.lib.call('rl_solve_pcg', ski.handle, dev_ptr(B), dev_ptr(X), k, float(tol), int(maxiter),
                 host_ptr(iters), host_ptr(resid), host_ptr(istop), ski.lib.stream_ptr(ski.device))
    return X, iters, resid, istop


def solve_pcg_lanczos(ski, B, tol=1e-4, maxiter=0, cap=1024):
    """solve_pcg that also returns each system's Lanczos matrix of the PRECONDITIONED operator
    (host (k, cap, 2): diagonal, off-diagonal) and r0^T P^-1 r0 (rl_solve_pcg_lanczos): what
    slq_quadratic_forms takes.  Returns (X, iterations, residuals, istop, lanczos, sqnorms)."""
    k = B.shape[0]
    X = torch.empty_like(B)
    iters = np.zeros(k, dtype=np.int32)
    istop = np.zeros(k, dtype=np.int32)
    resid = np.zeros(k, dtype=np.float64)
    lanczos = np.zeros((k, int(cap), 2), dtype=np.float64)
    sq = np.zeros(k, dtype=np.float64)
    if k == 0:
        return X, iters, resid, istop, lanczos, sq
    ski.lib.call('rl_solve_pcg_lanczos', ski.handle, dev_ptr(B), dev_ptr(X), k, float(tol), int(maxiter),
                 host_ptr(iters), host_ptr(resid), host_ptr(istop), host_ptr(lanczos), int(cap),
                 host_ptr(sq), ski.lib.stream_ptr(ski.device))
    return X, iters, resid, istop, lanczos, sq


def slq_quadratic_forms(lanczos, iters, sqnorms, lib=None):
    """r^T log(K) r for each system from its Lanczos tridiagonal (Gauss
    quadrature): ||r||^2 * sum_j tau_j^2 log(theta_j), (theta, first
    eigenvector components tau) the eigenpairs of T_k.  The library's host helper
    (rl_slq_log_quadrature: implicit QL carrying one eigenvector row, O(k^2) per system, the
    systems over the host's cores) -- LAPACK through SciPy returns whole eigenvector matrices:
    16 ms per system at the 420 steps of a C5 solve, 2 s for its 128 probes."""
    import os
    lib = lib or _lib.get_library()
    lanczos = np.ascontiguousarray(lanczos, dtype=np.float64)
    k = lanczos.shape[0]
    out = np.zeros(k)
    if k == 0:
        return out
    its = np.ascontiguousarray(np.asarray(iters), dtype=np.int32)
    sq = np.ascontiguousarray(np.asarray(sqnorms), dtype=np.float64)
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    lib.call('rl_slq_log_quadrature', host_ptr(lanczos), int(k), int(lanczos.shape[1]),
             host_ptr(its), host_ptr(sq), host_ptr(out), int(max(1, min(cores, 32))))
    failed = np.flatnonzero(np.isnan(out))
    if len(failed):          # (an iteration that did not settle: LAPACK takes those systems)
        out[failed] = slq_quadratic_forms_scipy(lanczos[failed], its[failed], sq[failed])
    return out


def slq_quadratic_forms_scipy(lanczos, iters, sqnorms):
    """The same through SciPy's eigh_tridiagonal (whole eigenvector matrices): the check of
    the library's helper in the tests."""
    from scipy.linalg import eigh_tridiagonal
    out = np.zeros(len(iters))
    for i, k in enumerate(iters):
        k = int(min(k, lanczos.shape[1]))
        if k < 1:
            continue
        d = lanczos[i, :k, 0]
        e = lanczos[i, :k - 1, 1]
        if k == 1:
            theta, tau2 = d[:1], np.ones(1)
        else:
            if not (np.all(np.isfinite(d)) and np.all(np.isfinite(e))):
                out[i] = np.nan          # (a recurrence that left the finite numbers)
                continue
            try:
                theta, vecs = eigh_tridiagonal(d, e)
            except np.linalg.LinAlgError:
                try:
                    theta, vecs = eigh_tridiagonal(d, e, lapack_driver='stev')
                except np.linalg.LinAlgError:
                    out[i] = np.nan
                    continue
            tau2 = vecs[0] ** 2
        keep = theta > 0
        out[i] = sqnorms[i] * np.sum(tau2[keep] * np.log(theta[keep]))
    return out


def cross_dots(lib, U, V, D, m):
    """out[v, a, b] = <U[v, a-th block], V[v, b-th block]> on the device."""
    k = U.shape[0]
    out = torch.empty((k, D, D), dtype=torch.float64, device=U.device)
    lib.call('rl_cross_dots', dev_ptr(U), dev_ptr(V), k, int(D), int(m),
             dev_ptr(out), lib.stream_ptr(U.device))
    return out


def segment_dots(lib, U, V, offsets_dev, D):
    """out[v, d] = sum over output d's slice of U[v] * V[v]."""
    k, n = U.shape
    out = torch.empty((k, D), dtype=torch.float64, device=U.device)
    lib.call('rl_segment_dots', dev_ptr(U), dev_ptr(V), dev_ptr(offsets_dev),
             k, int(n), int(D), dev_ptr(out), lib.stream_ptr(U.device))
    return out
