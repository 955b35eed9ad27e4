"""GPU parity at BASELINE.json's FULL sizes (pytest -m gpu, through the C ABI).

C5 = synthetic D=10, Q=5, m=10^5 (grid 100 004, n = 10^6, 128 probes + y): the
instantiations only this size reaches -- the 400 x 512 split with its 80+ KB
row tiles, chunked products on two streams (>= 8 MB of intermediates per pair),
the LDS-staged W^T / W products (>= 2^22 grid entries per batch), the
long-system loops of the solver's vector kernels at 129 systems -- are hit by
SIZE here, never by an environment knob.  C2 = D=4, Q=3, m=5000, 16 probes + y.

Reference semantics: runlmc/linalg/bttb.py:144-148, kronecker.py:39-46,
approx/ski.py:13-16, approx/iterative.py:23-62.  Tolerances: products 1e-11
relative to max|y| (fp64 FFT roundoff, SURVEY 8c states 1e-10); MINRES iterates
see CAP below.
"""
import numpy as np
import pytest
import torch

from oracle import operators as ops
from oracle import likelihood as olik
from oracle.kernels import KernelSpec, RBFSpec
from oracle.solver import iterative_solve, minres_ps

pytestmark = pytest.mark.gpu

REL = 1e-11
# Fixed-count MINRES comparisons.  K~ = (fast-decaying RBF spectrum) + noise is
# numerically rank-revealing: after ~10 Lanczos steps the next vector is
# determined by roundoff (beta_k at the noise floor), so from there on two
# correct implementations agree only to the accuracy of the iterate itself
# (measured on the emulator at C5: 8e-13 after 8 iterations, O(1) after 16).
# At C2 the floor is reached sooner: 1.2e-9 after 8 iterations on the GPU.
# The recurrences are therefore compared after CAP iterations at 1e-8 and the
# converged iterates at the size of the last updates.
CAP = 6


@pytest.fixture(scope='module')
def native():
    from runlmc_amd import _lib
    lib = _lib.use_library(None) or _lib.get_library()
    assert lib.is_hip, 'GPU tests must run against the HIP build'
    return lib


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _spec(p):
    spec = KernelSpec(p.D, [RBFSpec(g) for g in p.inv_lengthscales], list(p.coreg_vecs),
                      list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    return spec


@pytest.fixture(scope='module')
def c5():
    from runlmc_amd.util import synth
    D, Q, R, m, npr = synth.CONFIGS['c5']
    return synth.make_problem(D, Q, R, m)


@pytest.fixture(scope='module')
def c5_gridop(native, c5):
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp
    g = GridOp(c5.D, c5.m, c5.Q)
    g.set_lmc(synth.tops(c5), list(c5.coreg_vecs), list(c5.coreg_diags))
    assert (g.L, g.N1, g.N2) == (204800, 400, 512)
    return g


@pytest.mark.parametrize('nvec,check', [(5, (0, 1, 2, 3, 4)), (129, (0, 1, 64, 127, 128))])
def test_c5_grid_mvm_vs_oracle(c5, c5_gridop, nvec, check):
    """The C5 grid operator (10, 5, 100 004) on 5 and on 129 vectors (the
    second spans several chunks of intermediates on two streams); a subset of
    the outputs against the oracle's 'sum' representation at 1e-11."""
    from runlmc_amd.util import synth
    g = c5_gridop
    gen = torch.Generator().manual_seed(17 + nvec)
    X = torch.randn(nvec, c5.D * c5.m, dtype=torch.float64, generator=gen)
    Y = g.mvm(X.to(g.device)).cpu().numpy()
    Bs = ops.coreg_mats(list(c5.coreg_vecs), list(c5.coreg_diags))
    toeps = [ops.BTTBOracle(t) for t in synth.tops(c5)]
    for v in check:
        ref = ops.grid_sum_matvec(Bs, toeps, X[v].numpy())
        assert _rel(Y[v], ref) < REL, v
    # batch position does not matter: the same vector alone gives the same bits
    # up to the pairing partner's roundoff (two vectors share a transform)
    alone = g.mvm(X[check[-1]:check[-1] + 1].to(g.device)).cpu().numpy()[0]
    assert _rel(alone, Y[check[-1]]) < 1e-13


def test_c5_both_forms_of_the_product(c5, c5_gridop):
    """The 129-vector C5 batch runs in the polynomial-subspace form (rank 24,
    accepted at set time); the same batch forced onto the transform kernels of the
    same handle agrees on EVERY vector to 1e-12 of the result's largest entry, and
    so do the single-top products the gradient uses (rl_gridop_mvm_top)."""
    g = c5_gridop
    rank, gate = g.form()
    assert rank == 24 and 129 * c5.D * c5.m >= gate
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(129, c5.D * c5.m, dtype=torch.float64, generator=gen).to(g.device)
    poly = g.mvm(X).cpu().numpy()
    poly_top = g.mvm(X, top=c5.Q - 1).cpu().numpy()
    g.set_form_gate(1 << 62)
    try:
        fft = g.mvm(X).cpu().numpy()
        fft_top = g.mvm(X, top=c5.Q - 1).cpu().numpy()
    finally:
        g.set_form_gate(-1)
    assert not np.array_equal(poly, fft)             # (two different kernels ran)
    scale = np.abs(fft).max(axis=1, keepdims=True)
    assert (np.abs(poly - fft) / scale).max() < 1e-12
    scale = np.abs(fft_top).max(axis=1, keepdims=True)
    assert (np.abs(poly_top - fft_top) / scale).max() < 1e-12


def test_c5_full_operator_vs_oracle(native, c5, c5_gridop):
    """K~ = W K_UU W^T + eps at n = 10^6 on a 129-vector batch (LDS-staged
    W^T / W products and sorted data order by size); three of the outputs
    against the oracle's operator in the representation the reference picks."""
    from runlmc_amd._native import SkiOp
    s = SkiOp(c5_gridop, c5.W, c5.WT)
    s.set_noise(c5.noise, c5.lens)
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(129, c5.n, dtype=torch.float64, generator=gen)
    Y = s.mvm(X.to(s.device)).cpu().numpy()
    op = olik.LMCOperatorOracle(_spec(c5), c5.grid_dists, c5.W, c5.WT, c5.lens)
    for v in (0, 63, 128):
        assert _rel(Y[v], op.matvec(X[v].numpy())) < REL, v
    # the two halves of the operator on their own
    G = s.apply_wt(X[:9].to(s.device)).cpu().numpy()
    assert _rel(G[8], c5.WT.dot(X[8].numpy())) < 1e-13
    Z = s.apply_w(torch.from_numpy(G).to(s.device)).cpu().numpy()
    assert _rel(Z[8], c5.W.dot(G[8])) < 1e-13


def test_c5_minres_converged_vs_oracle(native):
    """alpha parity AT SCALE: the C5 system with eps = 1 and a noise floor
    (condition number of a few hundred, so the Krylov iterates are insensitive
    to roundoff), all 129 right-hand sides on the device; y and one probe against
    the oracle's MINRES iterate at 1e-6.  (The reference's ABSOLUTE 1e-4 residual
    is 1.7e-7 relative at n = 10^6: SciPy's own rtol = 1e-10 / Acond exits end
    both sides first, at the same iteration and the same residual.)"""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd._native import solve_batch
    D, Q, R, m, npr = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m, eps=1.0)
    p.noise = p.noise + 20.0
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    dop = K.device_operator()
    rng = np.random.RandomState(11)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(npr)])
    Bd = torch.from_numpy(B).to(dop.device)
    oop = olik.LMCOperatorOracle(_spec(p), p.grid_dists, p.W, p.WT, p.lens)
    # (a) the same NUMBER of iterations on both sides: iterate parity proper
    Xc, itc = solve_batch(dop, Bd, tol=1e-4, maxiter=CAP)[:2]
    Xc = Xc.cpu().numpy()
    assert np.all(np.array(itc) == CAP)
    for v in (0, 77):
        xo, info = minres_ps(oop.matvec, B[v], rtol=1e-10, maxiter=CAP)[:2]
        assert _rel(Xc[v], xo) < 1e-8, v
    # (b) run to the reference's stopping rule: same exit, same count (the
    # rtol = 1e-10 test is crossed at the roundoff floor, so the count may move
    # by a few iterations and the iterates by the size of the last updates)
    X, it, rs, st = solve_batch(dop, Bd, tol=1e-4)[:4]
    X = X.cpu().numpy()
    rs, it = np.array(rs), np.array(it)
    for v in (0, 77):
        xo, ito, erro, ok = iterative_solve(oop.matvec, B[v], tol=1e-4)
        assert abs(int(it[v]) - ito) <= max(3, ito // 10), (v, it[v], ito)
        assert _rel(X[v], xo) < 2e-4, v
        # the reported residual is the true one, through the ORACLE's operator,
        # and as small as the reference's own
        true = np.linalg.norm(B[v] - oop.matvec(X[v]))
        assert abs(true - rs[v]) <= 1e-9 + 1e-6 * true
        assert rs[v] <= max(1e-4, 1.5 * erro)
    assert rs.max() <= 3 * rs[[0, 77]].max()


def test_c2_converged_alpha_vs_oracle(native):
    """The same at C2 (D=4, Q=3, m=5000, 16 probes + y): converged device
    solves (17 systems: the fused small-system rounds) against the oracle's
    MINRES iterate (1e-8 after CAP iterations, 2e-4 at the reference's exit)."""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd._native import solve_batch
    D, Q, R, m, npr = synth.CONFIGS['c2']
    p = synth.make_problem(D, Q, R, m, eps=1.0)
    p.noise = p.noise + 2.0
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    dop = K.device_operator()
    rng = np.random.RandomState(12)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(npr)])
    Bd = torch.from_numpy(B).to(dop.device)
    oop = olik.LMCOperatorOracle(_spec(p), p.grid_dists, p.W, p.WT, p.lens)
    Xc, itc = solve_batch(dop, Bd, tol=1e-4, maxiter=CAP)[:2]
    Xc = Xc.cpu().numpy()
    assert np.all(np.array(itc) == CAP)
    for v in (0, 1, 16):
        xo = minres_ps(oop.matvec, B[v], rtol=1e-10, maxiter=CAP)[0]
        assert _rel(Xc[v], xo) < 1e-8, v
    X, it, rs, st = solve_batch(dop, Bd, tol=1e-4)[:4]
    X = X.cpu().numpy()
    rs = np.array(rs)
    for v in (0, 1, 16):
        xo, ito, erro, ok = iterative_solve(oop.matvec, B[v], tol=1e-4)
        assert abs(int(it[v]) - ito) <= max(3, ito // 10), (v, it[v], ito)
        assert _rel(X[v], xo) < 2e-4, v
        assert rs[v] <= max(1e-4, 1.5 * erro)


def test_c5_gradient_terms_vs_oracle(native, c5, c5_gridop):
    """One Gram matrix of the batched gradient at C5's shape: P_T(u, v)[a, b] =
    (W^T u)_a . T_q (W^T v)_b from the device (single-top product + cross dots)
    against NumPy on the oracle's Toeplitz product."""
    from runlmc_amd.util import synth
    from runlmc_amd._native import cross_dots
    g = c5_gridop
    rng = np.random.RandomState(3)
    U = rng.randn(2, c5.D * c5.m)
    q = c5.Q - 1
    Ud = torch.from_numpy(U).to(g.device)
    TU = g.mvm(Ud, top=q)
    P = cross_dots(g.lib, Ud, TU, c5.D, c5.m).cpu().numpy()
    toep = ops.BTTBOracle(synth.tops(c5)[q])
    for v in range(2):
        u = U[v].reshape(c5.D, c5.m)
        tu = np.array([toep.matvec(r) for r in u])
        ref = u @ tu.T
        assert np.abs(P[v].reshape(c5.D, c5.D) - ref).max() < 1e-10 * np.abs(ref).max()


def test_c5_gradient_at_fixed_iterations_both_forms(native, c5):
    """C5 at its REAL noise level (eps = 0.1, the reference's default,
    benchmarks/benchlib/bench.py:114-115): the reference's rule stops these solves
    unconverged (SciPy's test1 at a residual of ~140, iteration counts that move
    with roundoff), so the end state pins nothing.  What is well defined is the
    state after a FIXED number of MINRES iterations (CAP, before the Lanczos
    vectors are roundoff-determined, see above): alpha, the probe solves and the
    complete gradient assembled from them must agree between the two forms of
    the grid product -- polynomial form and transform kernels of the same handle
    -- to 1e-7 of the gradient's norm, and alpha with the oracle's MINRES."""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDeriv
    from runlmc_amd._native import solve_batch
    p = c5
    fk = synth.functional_kernel(p)
    ad = (0,)
    rng = np.random.RandomState(4321)
    nprobe = 16
    rs = rng.randint(0, 2, (nprobe, p.n)) * 2.0 - 1
    B = np.vstack([p.y, rs])

    class Fixed:
        def __init__(self, X, dev):
            self.X, self.dev = X, dev

        def generate(self, K, y, rs=None):
            t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(self.dev)
            return StochasticDeriv(self.X[0], t(B[1:]), self.X[1:], nprobe)

    grads, alphas = {}, {}
    for form in ('poly', 'fft'):
        K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        op = K.device_operator()
        assert op.grid.form()[0] in (24, 32, 48)
        op.grid.set_form_gate(0 if form == 'poly' else 1 << 62)
        X, it = solve_batch(op, torch.from_numpy(B).to(op.device), tol=1e-4, maxiter=CAP)[:2]
        assert np.all(np.array(it) == CAP)
        lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys,
                                  Fixed(X, op.device))
        g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(), lik.kernel_gradients(),
             lik.noise_gradient())
        grads[form] = np.concatenate([np.ravel(x) for x in g[0] + g[1] + [np.hstack(g[2])] + [g[3]]])
        alphas[form] = X[0].cpu().numpy()
    gn = np.linalg.norm(grads['fft'])
    diff = np.linalg.norm(grads['poly'] - grads['fft']) / gn
    print('C5 eps=0.1, %d iterations: gradient norm %.4g, forms differ by %.3g of it; alpha by %.3g'
          % (CAP, gn, diff, _rel(alphas['poly'], alphas['fft'])))
    assert diff < 1e-7
    assert _rel(alphas['poly'], alphas['fft']) < 1e-8
    oop = olik.LMCOperatorOracle(_spec(p), p.grid_dists, p.W, p.WT, p.lens)
    xo = minres_ps(oop.matvec, B[0], rtol=1e-10, maxiter=CAP)[0]
    assert _rel(alphas['poly'], xo) < 1e-8


def test_polynomial_gate_boundary_full_size():
    """The acceptance gate at its boundary on the C5 grid (100 004 points): see
    parity_suite.check_polynomial_gate_boundary."""
    import parity_suite as ps
    rep = ps.check_polynomial_gate_boundary(m=100004, factor=1.5)
    print('last accepted parameters per rank:', rep)
