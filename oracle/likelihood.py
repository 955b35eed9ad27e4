"""Oracle: LMC operator assembly, probe solves and the Hutchinson gradient.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Scope: one active-dimension set; 1-D grids (the reference's C1-C5 workloads)
and N-D grids (BTTB kernels).
"""
import numpy as np
import scipy.linalg as la

from . import operators as ops
from .solver import iterative_solve


def choose_ktype(spec, active_dim=(0,)):
    """Representation the reference auto-selects (reference
    runlmc/lmc/grid_kernel.py:52-64)."""
    if spec.Q == 1:
        return 'sum'
    tot_rank = spec.total_rank(active_dim)
    no_diag = (not spec.num_lmc[active_dim]) and (not spec.num_indep[active_dim])
    correction = spec.D if no_diag else 0
    return 'slfm' if tot_rank + spec.D < spec.D ** 2 + correction else 'bt'


def slfm_identity_terms(spec, active_dim=(0,)):
    """How many IDENTITY matrices the reference's 'slfm' representation puts on
    the grid in the place of parts the model does not have (reference
    grid_kernel.py:87-88: no coregionalised kernel in the set -> Identity(m)
    instead of the coregionalised part; :104-105: neither LMC nor independent
    kernels -> Identity(m) instead of the diagonal part).  Pinned by
    tests/golden/slfm_quirk.npz (the reference's own GridKernel)."""
    kidxs = spec.active_dims[active_dim]
    count = 0
    if not spec.filter_non_indep_idxs(kidxs):
        count += 1
    if spec.num_lmc[active_dim] == 0 and spec.num_indep[active_dim] == 0:
        count += 1
    return count


class LMCOperatorOracle:
    """K~ = W K_UU W^T + diag(eps) in one of the three representations
    (reference grid_kernel.py:22-74).  `reference_slfm_identity=True` reproduces
    the 'slfm' representation's identity terms (slfm_identity_terms) instead of
    the mathematical sum_q B_q (x) K_q."""

    def __init__(self, spec, grid_dists, W, WT, lens, ktype=None,
                 active_dim=(0,), reference_slfm_identity=False):
        self._eye = (slfm_identity_terms(spec, active_dim)
                     if reference_slfm_identity else 0)
        self.spec = spec
        self.W, self.WT = W, WT
        self.lens = list(lens)
        self.n = W.shape[0]
        self.sizes = tuple(np.shape(grid_dists))
        self.m = int(np.prod(self.sizes))
        self.ktype = ktype or choose_ktype(spec, active_dim)
        self.tops = np.asarray(spec.eval_kernels_fixed_dim(grid_dists, active_dim)
                               ).reshape(-1, self.m)
        self.Bs = spec.coreg_mats(active_dim)
        self.noise_diag = np.repeat(spec.noise, self.lens)
        if self.ktype == 'sum':
            self._toeps = [ops.BTTBOracle(t, self.sizes) for t in self.tops]

    def grid_matvec(self, g):
        if self.ktype == 'sum':
            return ops.grid_sum_matvec(self.Bs, self._toeps, g)
        if self.ktype == 'bt':
            return ops.grid_bt_matvec(self.Bs, self.tops, self.sizes, g)
        if self.ktype == 'slfm':
            out = ops.grid_slfm_matvec(self.spec.coreg_vecs,
                                       self.spec.coreg_diags, self.tops,
                                       self.sizes, g)
            return out + self._eye * np.asarray(g, dtype=np.float64)
        raise AssertionError(self.ktype)

    def matvec(self, x):
        return ops.full_matvec(self.W, self.WT, self.grid_matvec,
                               self.noise_diag, x)

    def as_numpy(self):
        return ops.dense_from_matvec(self.matvec, self.n)


def draw_probes(n_it, n, rng=None):
    """Rademacher probes exactly as the reference draws them (reference
    runlmc/lmc/stochastic_deriv.py:35): legacy global RNG unless an explicit
    RandomState is given."""
    r = np.random if rng is None else rng
    return r.randint(0, 2, (n_it, n)) * 2 - 1


def solve_all(op, y, rs, tol=1e-4, minres=True):
    """alpha and K~^{-1} r_i by N+1 independent solves (reference
    stochastic_deriv.py:39-52).  Returns (alpha, inv_rs, iters, errs)."""
    sols, iters, errs = [], [], []
    for rhs in [y] + [r.astype(np.float64) for r in rs]:
        x, it, err, _ = iterative_solve(op.matvec, rhs, tol=tol, minres=minres)
        sols.append(x)
        iters.append(it)
        errs.append(err)
    return sols[0], np.array(sols[1:]), np.array(iters), np.array(errs)


def _half_quad_minus_trace(dK_mv, alpha, rs, inv_rs):
    """0.5 (alpha^T dK alpha - (1/N) sum_i (K^-1 r_i)^T dK r_i) (reference
    runlmc/lmc/derivative.py:5-6, stochastic_deriv.py:69-78)."""
    quad = alpha.dot(dK_mv(alpha))
    tr = 0.0
    for r, rinv in zip(rs, inv_rs):
        tr += rinv.dot(dK_mv(r.astype(np.float64)))
    return 0.5 * (quad - tr / len(rs))


def stochastic_gradients(spec, grid_dists, W, WT, lens, alpha, rs, inv_rs,
                         active_dim=(0,)):
    """The four gradient families, one dK operator and N+1 MVMs per
    hyper-parameter, exactly the reference's loops (reference
    runlmc/lmc/likelihood.py:48-96,112-131).

    Returns dict(coreg_vec=[(R_q x D)]*Q, coreg_diag=[(D,)]*Q,
                 kernel=[[p_q floats]]*Q, noise=(D,))."""
    D, Q = spec.D, spec.Q
    sizes = tuple(np.shape(grid_dists))
    dists = {active_dim: grid_dists}
    mats = [ops.BTTBOracle(np.ravel(k), sizes) for k in spec.eval_kernels(dists)]
    dmats = spec.eval_kernel_gradients(dists)

    def ski_kron(B, toep):
        return lambda x: W.dot(ops.kron_matvec(B, toep, WT.dot(x)))

    def deriv(mv):
        return _half_quad_minus_trace(mv, alpha, rs, inv_rs)

    g_vec = []
    for q, a in enumerate(spec.coreg_vecs):
        g = np.zeros(a.shape)
        for i, ai in enumerate(a):
            for j in range(D):
                dB = np.zeros((D, D))
                dB[j] += ai
                dB.T[j] += ai
                g[i, j] = deriv(ski_kron(dB, mats[q]))
        g_vec.append(g)

    g_diag = []
    for q in range(Q):
        g = np.zeros(D)
        for i in range(D):
            dB = np.zeros((D, D))
            dB[i, i] = 1
            g[i] = deriv(ski_kron(dB, mats[q]))
        g_diag.append(g)

    g_kern = []
    for q, B in enumerate(spec.coreg_mats()):
        g_kern.append([deriv(ski_kron(B, ops.BTTBOracle(np.ravel(dk), sizes)))
                       for dk in dmats[q]])

    g_noise = np.zeros(D)
    for d in range(D):
        e = np.zeros(D)
        e[d] = 1
        mask = np.repeat(e, lens)
        g_noise[d] = deriv(lambda x, mask=mask: mask * x)

    return dict(coreg_vec=g_vec, coreg_diag=g_diag, kernel=g_kern,
                noise=g_noise)


# --- dense exact twin (reference likelihood.py:137-217, exact_deriv.py) ----

def exact_kernel_dense(spec, Xs):
    """Exact (non-SKI) LMC covariance on the data (reference
    likelihood.py:138-153): sum_q B_q[d(i), d(j)] k_q(|x_i - x_j|) + noise."""
    lens = [len(X) for X in Xs]
    x = np.concatenate([np.asarray(X, dtype=np.float64).ravel() for X in Xs])
    dist = np.abs(x[:, None] - x[None, :])
    out_of = np.repeat(np.arange(spec.D), lens)
    K = np.zeros((len(x), len(x)))
    for B, k in zip(spec.coreg_mats(), spec._kernels):
        K += B[np.ix_(out_of, out_of)] * k.from_dist(dist)
    K += np.diag(np.repeat(spec.noise, lens))
    return K


def logdet_dense(K):
    """2 sum log diag chol (reference
    runlmc/models/interpolated_llgp.py:262-276)."""
    c = la.cho_factor(K)[0]
    return 2.0 * np.sum(np.log(np.diag(c)))


def exact_gradients_from_dense(K, y, dK_list):
    """0.5 (alpha^T dK alpha - tr(K^-1 dK)) for dense dK (reference
    exact_deriv.py:13-23)."""
    c = la.cho_factor(K)
    Kinv = la.cho_solve(c, np.identity(K.shape[0]))
    alpha = la.cho_solve(c, y)
    return [0.5 * (alpha.dot(dK.dot(alpha)) - (dK * Kinv).sum())
            for dK in dK_list]


def exact_gradients(spec, Xs, y):
    """The four gradient families of the EXACT dense likelihood -- the twin the reference's
    `bench.py opt` holds the approximate gradients against (reference
    runlmc/lmc/likelihood.py:48-96 loops over likelihood.py:137-217 ExactLMCLikelihood,
    with exact_deriv.py:13-23:  dL/dt = 0.5 (alpha^T dK alpha - tr(K^-1 dK)) ).

    Restated with  M = alpha alpha^T - K^-1:  dL/dt = 0.5 sum_ij M_ij dK_ij, and every dK of
    the loops is  dB[d(i), d(j)] Kq[i, j]  (likelihood.py:197-206), so one D x D block sum
    S_q[a, b] = sum_{i in a, j in b} M_ij Kq_ij  per kernel matrix gives all of its
    derivatives:  0.5 sum_ab dB[a, b] S_q[a, b].  Returns (dict like stochastic_gradients,
    alpha, K)."""
    D, Q = spec.D, spec.Q
    lens = [len(X) for X in Xs]
    x = np.concatenate([np.asarray(X, dtype=np.float64).ravel() for X in Xs])
    dist = np.abs(x[:, None] - x[None, :])
    K = exact_kernel_dense(spec, Xs)
    c = la.cho_factor(K)
    M = -la.cho_solve(c, np.identity(K.shape[0]))
    alpha = la.cho_solve(c, y)
    M += np.outer(alpha, alpha)
    ends = np.cumsum(lens)
    begins = ends - np.asarray(lens)

    def block_sums(Kq):
        P = M * Kq
        S = np.zeros((D, D))
        for a in range(D):
            for b in range(D):
                S[a, b] = P[begins[a]:ends[a], begins[b]:ends[b]].sum()
        return S

    g_vec, g_diag, g_kern = [], [], []
    dists = {(0,): dist}
    grads = spec.eval_kernel_gradients(dists)
    for q, (a_q, B, k) in enumerate(zip(spec.coreg_vecs, spec.coreg_mats(), spec._kernels)):
        S = block_sums(k.from_dist(dist))
        g = np.zeros(np.shape(a_q))
        for i, ai in enumerate(np.atleast_2d(a_q)):
            for j in range(D):
                # dB = e_j a_i^T + a_i e_j^T  (likelihood.py:52-57)
                g[i, j] = 0.5 * (ai.dot(S[j, :]) + ai.dot(S[:, j]))
        g_vec.append(g)
        g_diag.append(0.5 * np.diag(S).copy())              # dB = e_i e_i^T (likelihood.py:68-71)
        g_kern.append([0.5 * np.sum(B * block_sums(dk)) for dk in grads[q]])   # likelihood.py:82-86
    g_noise = np.array([0.5 * np.trace(M[b:e, b:e]) for b, e in zip(begins, ends)])  # :92-95
    return dict(coreg_vec=g_vec, coreg_diag=g_diag, kernel=g_kern, noise=g_noise), alpha, K
