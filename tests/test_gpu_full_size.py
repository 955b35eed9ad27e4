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
        assert op.grid.form()[0] in (24, 32, 36, 40, 48)
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


# ---------------------------------------------------------------------------
# The reference benchmark's other kernel families (benchmarks/benchlib/bench.py:94,
# 284-297; kern/std_periodic.py:44-67, kern/matern32.py:40-55) at the shapes
# bench.py times them at: the instantiations and multi-tile paths only these
# sizes reach (k_sf_apply<NS, 10> / <NS, 4>, persistent workgroups that draw
# many tiles each, rank-48 projection / expansion on 1290 rows, a rank-48
# polynomial part accumulating onto a filter part at D >= 8).

FAMILY_FORMS = {'periodic': ['polynomial'] * 5, 'matern': ['filter'] * 5,
                'mix': ['polynomial', 'polynomial', 'filter', 'polynomial', 'polynomial']}
FORM_NAMES = {0: 'transform', 1: 'polynomial', 2: 'filter'}


def _family_gridop(D, Q, m_data, kern):
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp
    p = synth.make_problem(D, Q, 1, m_data, kern=kern)
    g = GridOp(D, p.m, Q)
    g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
    return p, g


def _oracle_rows(p, X, rows):
    from runlmc_amd.util import synth
    Bs = ops.coreg_mats(list(p.coreg_vecs), list(p.coreg_diags))
    toeps = [ops.BTTBOracle(t) for t in synth.tops(p)]
    return {v: ops.grid_sum_matvec(Bs, toeps, X[v].numpy()) for v in rows}


@pytest.mark.parametrize('kern', ['periodic', 'matern', 'mix'])
def test_c5_family_product_vs_oracle(native, kern):
    """C5 (D=10, Q=5, grid 100 004), 129 vectors, the forms bench.py reports for
    the family: five of the outputs against the oracle's 'sum' representation at
    1e-11, ALL 129 against the transform kernels of the same handle at 1e-12.
    1290 rows: 196 chunks x 129 vectors = 25 284 tiles for k_sf_apply's 512
    persistent workgroups (about 49 tiles each)."""
    p, g = _family_gridop(10, 5, 100000, kern)
    forms, structured = g.top_forms()
    assert [FORM_NAMES[f] for f in forms] == FAMILY_FORMS[kern] and structured
    rank, gate = g.form()
    # (rl_gridop_form reports a rank only when EVERY top is in the polynomial form)
    assert (rank in (36, 40, 48) if kern == 'periodic' else rank == 0) and 129 * p.D * p.m >= gate
    gen = torch.Generator().manual_seed(23)
    X = torch.randn(129, p.D * p.m, dtype=torch.float64, generator=gen)
    X[128] = torch.cos(7 * torch.linspace(0, 1, p.D * p.m, dtype=torch.float64)) + 0.5   # coherent
    Xd = X.to(g.device)
    got = g.mvm(Xd).cpu().numpy()
    for v, ref in _oracle_rows(p, X, (0, 1, 64, 127, 128)).items():
        assert _rel(got[v], ref) < REL, (kern, v)
    g.set_form_gate(1 << 62)
    try:
        fft = g.mvm(Xd).cpu().numpy()
    finally:
        g.set_form_gate(-1)
    assert not np.array_equal(got, fft)
    scale = np.abs(fft).max(axis=1, keepdims=True)
    assert (np.abs(got - fft) / scale).max() < 1e-12
    # the product does not depend on who draws which tile: a second run gives the same bits
    assert np.array_equal(got, g.mvm(Xd).cpu().numpy())


def test_c5_single_top_products_matern_and_derivative(native):
    """The gradient's dK products at C5 size: a Matern-3/2 row (two states per
    direction) and its d/d gamma row (-3 gamma r^2 exp(-sqrt(3) gamma r), three
    states; reference kern/matern32.py:50-55) as single-top products over 17
    and 129 vectors, against the oracle's Toeplitz product and the transform
    kernels."""
    from runlmc_amd._native import GridOp
    from runlmc_amd.util import synth
    D = 10
    p = synth.make_problem(D, 5, 1, 100000, kern='matern')
    kobj = synth.kernel_objects(p.kern_desc)[2]
    tops = np.array([kobj.from_dist(p.grid_dists), kobj.kernel_gradient(p.grid_dists)[0]])
    g = GridOp(D, p.m, 2)
    g.set_lmc(tops, [None, None], [np.zeros(D)] * 2)
    assert g.top_forms()[0] == [2, 2]
    gen = torch.Generator().manual_seed(29)
    X = torch.randn(129, D * p.m, dtype=torch.float64, generator=gen)
    Xd = X.to(g.device)
    for t in range(2):
        toep = ops.BTTBOracle(tops[t])
        for nvec in (17, 129):
            got = g.mvm(Xd[:nvec], top=t).cpu().numpy()
            for v in (0, nvec - 1):
                ref = np.concatenate([toep.matvec(r) for r in X[v].numpy().reshape(D, p.m)])
                assert _rel(got[v], ref) < REL, (t, nvec, v)
        g.set_form_gate(1 << 62)
        try:
            fft = g.mvm(Xd, top=t).cpu().numpy()
        finally:
            g.set_form_gate(-1)
        got = g.mvm(Xd, top=t).cpu().numpy()
        assert (np.abs(got - fft) / np.abs(fft).max(axis=1, keepdims=True)).max() < 1e-12


@pytest.mark.parametrize('kern', ['periodic', 'matern', 'mix'])
@pytest.mark.parametrize('nvec', [17, 1024])
def test_c2_family_product_vs_oracle(native, kern, nvec):
    """The C2 shape (D=4, Q=3, grid 5004) with the batch gate lifted -- the D = 4
    instantiations -- at the solver's 17 vectors and at 1024 vectors (10 chunks x
    1024 = 10 240 tiles): oracle on four vectors at 1e-11, the transform kernels on
    all of them at 1e-12.  ('mix' at D < 8 with a rank-48 top is handed to the
    transform kernels by forms_setup; the single-top products keep their forms.)"""
    p, g = _family_gridop(4, 3, 5000, kern)
    forms, structured = g.top_forms()
    want = {'periodic': [1, 1, 1], 'matern': [2, 2, 2], 'mix': [1, 1, 2]}[kern]
    assert forms == want, forms
    gen = torch.Generator().manual_seed(31 + nvec)
    X = torch.randn(nvec, p.D * p.m, dtype=torch.float64, generator=gen)
    Xd = X.to(g.device)
    g.set_form_gate(0)
    try:
        got = g.mvm(Xd).cpu().numpy()
        tops_got = [g.mvm(Xd, top=q).cpu().numpy() for q in range(p.Q)]
    finally:
        g.set_form_gate(-1)
    for v, ref in _oracle_rows(p, X, (0, 1, nvec // 2, nvec - 1)).items():
        assert _rel(got[v], ref) < REL, (kern, v)
    g.set_form_gate(1 << 62)
    try:
        fft = g.mvm(Xd).cpu().numpy()
        tops_fft = [g.mvm(Xd, top=q).cpu().numpy() for q in range(p.Q)]
    finally:
        g.set_form_gate(-1)
    assert (np.abs(got - fft) / np.abs(fft).max(axis=1, keepdims=True)).max() < 1e-12
    for a, b in zip(tops_got, tops_fft):
        assert (np.abs(a - b) / np.abs(b).max(axis=1, keepdims=True)).max() < 1e-12
    if structured:
        assert not np.array_equal(got, fft)


@pytest.mark.parametrize('kern', ['matern', 'mix'])
def test_c5_family_gradient_at_fixed_iterations(native, kern):
    """As test_c5_gradient_at_fixed_iterations_both_forms for the Matern and mix
    families: after CAP MINRES iterations the complete gradient (all four
    families of partial derivatives, kernel_gradients through the filter form's
    three-state derivative rows) from the structured forms and from the transform
    kernels of the same handle agree to 1e-7 of its norm, alpha to 1e-8 and with
    the oracle's MINRES."""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDeriv
    from runlmc_amd._native import solve_batch
    from oracle.kernels import Matern32Spec, StdPeriodicSpec
    p = synth.make_problem(10, 5, 1, 100000, kern=kern)
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
    for form in ('structured', 'fft'):
        K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        op = K.device_operator()
        assert op.grid.top_forms()[1]
        op.grid.set_form_gate(0 if form == 'structured' else 1 << 62)
        X, it = solve_batch(op, torch.from_numpy(B).to(op.device), tol=1e-4, maxiter=CAP)[:2]
        assert np.all(np.array(it) == CAP)
        lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys,
                                  Fixed(X, op.device))
        g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(), lik.kernel_gradients(),
             lik.noise_gradient())
        grads[form] = np.concatenate([np.ravel(x) for x in g[0] + g[1] + [np.hstack(g[2])] + [g[3]]])
        alphas[form] = X[0].cpu().numpy()
    gn = np.linalg.norm(grads['fft'])
    diff = np.linalg.norm(grads['structured'] - grads['fft']) / gn
    print('C5 %s, %d iterations: gradient norm %.4g, forms differ by %.3g of it; alpha by %.3g'
          % (kern, CAP, gn, diff, _rel(alphas['structured'], alphas['fft'])))
    assert diff < 1e-7
    assert _rel(alphas['structured'], alphas['fft']) < 1e-8
    make = {'rbf': RBFSpec, 'periodic': StdPeriodicSpec, 'matern': Matern32Spec}
    spec = KernelSpec(p.D, [make[d[0]](*d[1:]) for d in p.kern_desc], list(p.coreg_vecs),
                      list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    xo = minres_ps(oop.matvec, B[0], rtol=1e-10, maxiter=CAP)[0]
    assert _rel(alphas['structured'], xo) < 1e-8


@pytest.mark.parametrize('kern', ['rbf', 'periodic'])
def test_c5_row_polynomial_rounds(native, kern, monkeypatch):
    """The row-polynomial form of the C5 operator (rl_rowpoly.h: K~ = F M F^T + eps, F = W Phi;
    k_rp_project on the fp64 matrix cores over 490 runs of 2048 rows and two blocks of
    vectors, k_rp_expand) -- what the solver's rounds run at this size: the 129-vector
    product against the oracle's operator on three vectors (1e-11) and against the same
    handle's interpolation-product path (RUNLMC_NO_RP) on all of them (1e-12); MINRES
    iterates after CAP iterations against that path (1e-8) and the oracle's MINRES; all 129
    systems with MINRES's vector update inside the projection (three vector blocks per
    workgroup there, k_minres2_bh) against the same rounds with B as its own kernel (1e-11)."""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd._native import solve_batch
    from oracle.kernels import StdPeriodicSpec
    p = synth.make_problem(10, 5, 1, 100000, kern=kern)
    fk = synth.functional_kernel(p)
    ad = (0,)
    gen = torch.Generator().manual_seed(41)
    X = torch.randn(129, p.n, dtype=torch.float64, generator=gen)
    out = {}
    for mode in ('rp', 'interp'):
        monkeypatch.delenv('RUNLMC_NO_RP', raising=False)
        if mode == 'interp':
            monkeypatch.setenv('RUNLMC_NO_RP', '1')
        K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        op = K.device_operator()
        assert op.grid.form()[0] == (24 if kern == 'rbf' else 36)
        Y = op.mvm(X.to(op.device)).cpu().numpy()
        Xs, it = solve_batch(op, X[:17].to(op.device), tol=1e-4, maxiter=CAP)[:2]
        assert np.all(np.array(it) == CAP)
        out[mode] = (Y, Xs.cpu().numpy())
        if mode == 'rp':
            # all 129 systems (k_rp_project with MINRES's vector update inside: three vector
            # blocks, the lone last vector) against the same rounds with B as its own kernel
            full = {}
            for fuse in (True, False):
                monkeypatch.delenv('RUNLMC_NO_RP_FUSE', raising=False)
                if not fuse:
                    monkeypatch.setenv('RUNLMC_NO_RP_FUSE', '1')
                    K2, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
                    op = K2.device_operator()       # (switches are read when a handle is made)
                Xa, ita = solve_batch(op, X.to(op.device), tol=1e-4, maxiter=CAP)[:2]
                assert np.all(np.array(ita) == CAP)
                full[fuse] = Xa.cpu().numpy()
            monkeypatch.delenv('RUNLMC_NO_RP_FUSE', raising=False)
            assert not np.array_equal(full[True], full[False])
            assert (np.abs(full[True] - full[False]) /
                    np.abs(full[False]).max(axis=1, keepdims=True)).max() < 1e-11
            assert (np.abs(full[True][:17] - out['rp'][1]) /
                    np.abs(out['rp'][1]).max(axis=1, keepdims=True)).max() < 1e-11
    Yr, Sr = out['rp']
    Yi, Si = out['interp']
    assert not np.array_equal(Yr, Yi)
    assert (np.abs(Yr - Yi) / np.abs(Yi).max(axis=1, keepdims=True)).max() < 1e-12
    assert (np.abs(Sr - Si) / np.abs(Si).max(axis=1, keepdims=True)).max() < 1e-8
    make = {'rbf': RBFSpec, 'periodic': StdPeriodicSpec}
    spec = KernelSpec(p.D, [make[d[0]](*d[1:]) for d in p.kern_desc], list(p.coreg_vecs),
                      list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    for v in (0, 80, 128):
        assert _rel(Yr[v], oop.matvec(X[v].numpy())) < REL, v
    xo = minres_ps(oop.matvec, X[0].numpy(), rtol=1e-10, maxiter=CAP)[0]
    assert _rel(Sr[0], xo) < 1e-8


def test_polynomial_gate_boundary_full_size():
    """The acceptance gate at its boundary on the C5 grid (100 004 points): see
    parity_suite.check_polynomial_gate_boundary."""
    import parity_suite as ps
    rep = ps.check_polynomial_gate_boundary(m=100004, factor=1.5)
    print('last accepted parameters per rank:', rep)


# --- round 6: direct solves through the polynomial form at full size ------------------------
def _dense_from_oracle(op, n, threads=16):
    """The oracle's dense K~ column block by column block (n = 20 000: 3.2 GB, a minute)."""
    Kd = np.empty((n, n))
    eye = np.zeros(n)
    for i in range(n):
        eye[i] = 1.0
        Kd[:, i] = op.matvec(eye)
        eye[i] = 0.0
    return 0.5 * (Kd + Kd.T)


@pytest.mark.parametrize('kern', ['rbf', 'periodic'])
def test_c5_direct_solve_reaches_the_reference_tolerance(native, kern):
    """C5, 128 probes + y: K~ = F M F^T + E inverted through the Woodbury identity
    (rl_solve_direct).  (1) Every system ends on the reference's rule ||b - K~ x|| < 1e-4
    (approx/iterative.py:36-42) -- which no C5 system reaches by MINRES in fp64 -- and the
    residual holds through an INDEPENDENT operator handle that runs on the transform kernels
    only (no polynomial form anywhere in it) and, for y and one probe, through the oracle's FFT
    operator; (2) refined to 1e-6 the same; (3) log det K~ from the determinant lemma against
    the stochastic Lanczos quadrature of a MINRES run carried to the stall of its residuals
    (3000 iterations), within 4 SEM of the quadrature; (4) alpha against that run's alpha at
    the level of its residual."""
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp, SkiOp, solve_direct, solve_batch, slq_quadratic_forms, MINRES_RULE
    from oracle.kernels import StdPeriodicSpec
    D, Q, R, m0, N = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m0, kern=kern)
    tops = synth.tops(p)
    g = GridOp(p.D, p.m, p.Q)
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    s = SkiOp(g, p.W, p.WT)
    s.set_noise(p.noise, p.lens)
    ok, logdet, cond = s.factor()
    assert ok, s.factor_reason
    rng = np.random.RandomState(4321)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])
    Bd = torch.from_numpy(B).to(s.device)
    g2 = GridOp(p.D, p.m, p.Q)
    g2.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    g2.set_form_gate(1 << 60)                       # transform kernels, whatever the batch
    s2 = SkiOp(g2, p.W, p.WT)
    s2.set_noise(p.noise, p.lens)
    kerns = ([RBFSpec(gm) for gm in p.inv_lengthscales] if kern == 'rbf' else
             synth.kernel_objects(p.kern_desc, rbf=RBFSpec, periodic=StdPeriodicSpec))
    spec = KernelSpec(p.D, kerns, list(p.coreg_vecs), list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    for tol in (1e-4, 1e-6):
        X, it, res, st = solve_direct(s, Bd, tol=tol)
        assert np.all(st == 10) and np.all(res < tol), (tol, res.max(), st)
        assert it.max() <= 3
        r2 = (Bd - s2.mvm(X)).norm(dim=1).cpu().numpy()
        assert np.all(r2 < 2 * tol), (tol, r2.max())
        assert np.all(np.abs(r2 - res) <= 0.5 * tol), (np.abs(r2 - res).max())
        Xh = X[:2].cpu().numpy()
        for i in range(2):
            ro = np.linalg.norm(B[i] - oop.matvec(Xh[i]))
            assert ro < 2 * tol, (tol, i, ro)
    # the Krylov run carried to the stall of its residuals: its quadrature and its alpha
    k = 33
    Xk, itk, resk, stk, lz = solve_batch(s, Bd[:k].contiguous(), MINRES_RULE, tol=1e-4, maxiter=3000,
                                         lanczos_cap=3000)
    est = slq_quadratic_forms(lz[1:], itk[1:], np.full(k - 1, float(p.n)))
    sem = est.std(ddof=1) / np.sqrt(len(est))
    assert abs(est.mean() - logdet) <= 4 * sem, (est.mean(), logdet, sem)
    da = float((X[0] - Xk[0]).norm() / X[0].norm())
    assert da < 1e-3, da          # (the stalled iterate carries a residual of ~3e-3 of ||b|| = 577)


def test_c2_direct_solve_vs_dense_oracle(native):
    """C2 (n = 20 000): alpha, four probe solves and log det K~ from the factorisation against
    the oracle's DENSE K~ (its FFT operator, column by column) and LAPACK's Cholesky: alpha at
    1e-8 of its largest entry after refinement to 1e-9, log det at 1e-10 relative -- the
    determinant lemma against the reference's own definition of the quantity
    (models/interpolated_llgp.py:262-276) at a BASELINE size."""
    import scipy.linalg as la
    from threadpoolctl import threadpool_limits
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp, SkiOp, solve_direct
    D, Q, R, m0, N = synth.CONFIGS['c2']
    p = synth.make_problem(D, Q, R, m0)
    g = GridOp(p.D, p.m, p.Q)
    g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
    s = SkiOp(g, p.W, p.WT)
    s.set_noise(p.noise, p.lens)
    ok, logdet, cond = s.factor()
    assert ok, s.factor_reason
    oop = olik.LMCOperatorOracle(_spec(p), p.grid_dists, p.W, p.WT, p.lens)
    Kd = _dense_from_oracle(oop, p.n)
    rng = np.random.RandomState(7)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(4)])
    with threadpool_limits(limits=16):
        cf = la.cho_factor(Kd, overwrite_a=True)
        ld_ref = 2.0 * np.log(np.diag(cf[0])).sum()
        Xref = la.cho_solve(cf, B.T).T
    assert abs(logdet - ld_ref) <= 1e-10 * abs(ld_ref), (logdet, ld_ref)
    X, it, res, st = solve_direct(s, torch.from_numpy(B).to(s.device), tol=1e-9)
    assert np.all(st == 10) and np.all(res < 1e-9), (res, st)
    for i in range(len(B)):
        assert _rel(X[i].cpu().numpy(), Xref[i]) < 1e-8, (i, _rel(X[i].cpu().numpy(), Xref[i]))


def test_c5_mix_preconditioned_cg_reaches_the_reference_tolerance(native):
    """C5, the 'mix' family (four smooth rows in the polynomial form and a Matern row on the
    filter kernels): the factorisation inverts the operator's projection on the polynomial
    subspace and serves as the M of preconditioned conjugate gradients (rl_solve_pcg; the
    reference: sla.cg(op, y, M=M), approx/iterative.py:47-51) -- on the first 96 functions of the
    larger basis (*available = 3, csrc/rl_solve.hip hz_try).  All 129 systems end on the
    reference's residual rule in < 20 iterations -- MINRES runs 590 to a residual of 178 -- and
    the residuals hold through an independent handle on the transform kernels and, for y,
    through the oracle's FFT operator."""
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp, SkiOp, solve_pcg
    from oracle.kernels import StdPeriodicSpec, Matern32Spec
    D, Q, R, m0, N = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m0, kern='mix')
    tops = synth.tops(p)
    g = GridOp(p.D, p.m, p.Q)
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    s = SkiOp(g, p.W, p.WT)
    s.set_noise(p.noise, p.lens)
    ok, _, _ = s.factor()
    assert ok and s.factor_mode == 3, (s.factor_mode, s.factor_reason)
    assert sorted(set(g.top_forms()[0])) == [1, 2]
    rng = np.random.RandomState(4321)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])
    Bd = torch.from_numpy(B).to(s.device)
    X, it, res, st = solve_pcg(s, Bd, tol=1e-4)
    assert np.all(st == 10) and np.all(res < 1e-4), (res.max(), sorted(set(st)))
    assert it.max() < 20, it.max()
    g2 = GridOp(p.D, p.m, p.Q)
    g2.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    g2.set_form_gate(1 << 60)
    s2 = SkiOp(g2, p.W, p.WT)
    s2.set_noise(p.noise, p.lens)
    r2 = (Bd - s2.mvm(X)).norm(dim=1).cpu().numpy()
    assert np.all(r2 < 1.5e-4), r2.max()
    spec = KernelSpec(p.D, synth.kernel_objects(p.kern_desc, rbf=RBFSpec, periodic=StdPeriodicSpec,
                                                matern=Matern32Spec),
                      list(p.coreg_vecs), list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    assert np.linalg.norm(B[0] - oop.matvec(X[0].cpu().numpy())) < 1.5e-4
    # the same operator preconditioned on its OWN basis (the polynomial rows' 36 functions:
    # *available = 2): more iterations to the same rule
    import os
    saved = {k: os.environ.get(k) for k in ('RUNLMC_DEBUG', 'RUNLMC_NO_PRECOND_HI_MIXED')}
    os.environ.update(RUNLMC_DEBUG='1', RUNLMC_NO_PRECOND_HI_MIXED='1')
    try:
        g3 = GridOp(p.D, p.m, p.Q)
        g3.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
        s3 = SkiOp(g3, p.W, p.WT)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    s3.set_noise(p.noise, p.lens)
    ok, _, _ = s3.factor()
    assert ok and s3.factor_mode == 2, (s3.factor_mode, s3.factor_reason)
    X3, it3, res3, st3 = solve_pcg(s3, Bd[:17].contiguous(), tol=1e-4)
    assert np.all(st3 == 10) and it.max() < it3.max() < 100, (it.max(), it3.max())
    assert ((X3 - X[:17]).norm(dim=1) / X[:17].norm(dim=1)).max().item() < 1e-4


def test_c5_matern_larger_basis_preconditioner(native):
    """C5, Matern rows only (no row in the polynomial form): the preconditioner is the Woodbury
    inverse on a basis of 192 polynomials per output (rl_ski_factor: *available = 3;
    csrc/rl_solve.hip hz_*), applied 48 columns at a time by the rank-48 kernels.  All 129 systems
    end on the reference's residual rule in < 100 iterations (1358 with the handle's 48 functions;
    MINRES exits at 1194 with a residual of 188), the residuals hold through an independent handle
    on the transform kernels and, for y, through the oracle's FFT operator; a parameter update
    rebuilds the map and the solve still converges."""
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp, SkiOp, solve_pcg
    from oracle.kernels import StdPeriodicSpec, Matern32Spec
    D, Q, R, m0, N = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m0, kern='matern')
    tops = synth.tops(p)
    g = GridOp(p.D, p.m, p.Q)
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    s = SkiOp(g, p.W, p.WT)
    s.set_noise(p.noise, p.lens)
    ok, _, _ = s.factor()
    assert ok and s.factor_mode == 3, (s.factor_mode, s.factor_reason)
    assert sorted(set(g.top_forms()[0])) == [2]
    rng = np.random.RandomState(4321)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])
    Bd = torch.from_numpy(B).to(s.device)
    X, it, res, st = solve_pcg(s, Bd, tol=1e-4)
    assert np.all(st == 10) and np.all(res < 1e-4), (res.max(), sorted(set(st)))
    assert it.max() < 100, it.max()
    g2 = GridOp(p.D, p.m, p.Q)
    g2.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    g2.set_form_gate(1 << 60)
    s2 = SkiOp(g2, p.W, p.WT)
    s2.set_noise(p.noise, p.lens)
    r2 = (Bd - s2.mvm(X)).norm(dim=1).cpu().numpy()
    assert np.all(r2 < 1.5e-4), r2.max()
    spec = KernelSpec(p.D, synth.kernel_objects(p.kern_desc, rbf=RBFSpec, periodic=StdPeriodicSpec,
                                                matern=Matern32Spec),
                      list(p.coreg_vecs), list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    assert np.linalg.norm(B[0] - oop.matvec(X[0].cpu().numpy())) < 1.5e-4
    # parameter update: couplings scaled; 17 systems (the small-batch projection)
    g.set_lmc(tops, [np.sqrt(1.5) * a for a in p.coreg_vecs], [1.5 * k for k in p.coreg_diags])
    g2.set_lmc(tops, [np.sqrt(1.5) * a for a in p.coreg_vecs], [1.5 * k for k in p.coreg_diags])
    X17, it17, res17, st17 = solve_pcg(s, Bd[:17].contiguous(), tol=1e-4)
    assert s.factor_mode == 3 or s.factor()[0] and s.factor_mode == 3
    assert np.all(st17 == 10) and it17.max() < 100, (it17.max(), sorted(set(st17)))
    r17 = (Bd[:17] - s2.mvm(X17)).norm(dim=1).cpu().numpy()
    assert np.all(r17 < 1.5e-4), r17.max()


def test_c2_preconditioned_logdet_vs_dense_oracle(native):
    """C2 size (n = 20 000), Matern rows: log det K~ on the preconditioned path -- log det P from the
    determinant lemma plus the preconditioned Lanczos quadrature of 16 extra conjugate-gradient
    solves (rl_ski_precond_sample, rl_solve_pcg_lanczos) -- against the oracle's DENSE K~ and
    LAPACK's Cholesky: within 4 standard errors, the standard error itself below 1e-3 of the value
    (the handle's 48 functions at this size; C5's larger basis: below 1e-5, next test)."""
    import scipy.linalg as la
    from threadpoolctl import threadpool_limits
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from oracle.kernels import StdPeriodicSpec, Matern32Spec
    D, Q, R, m0, N = synth.CONFIGS['c2']
    p = synth.make_problem(D, Q, R, m0, kern='matern')
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    M = K.preconditioner
    assert M is not None and not M.exact
    est, sem, it = M.logdet_estimate(tol=1e-6)
    spec = KernelSpec(p.D, synth.kernel_objects(p.kern_desc, rbf=RBFSpec, periodic=StdPeriodicSpec,
                                                matern=Matern32Spec),
                      list(p.coreg_vecs), list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    Kd = _dense_from_oracle(oop, p.n)
    with threadpool_limits(limits=16):
        ld_ref = 2.0 * np.log(np.diag(la.cholesky(Kd, lower=True, overwrite_a=True))).sum()
    assert abs(est - ld_ref) <= 4.0 * sem + 1e-9 * abs(ld_ref), (est, ld_ref, sem)
    assert sem < 1e-3 * abs(ld_ref), (sem, ld_ref)


def test_c5_preconditioned_logdet_is_consistent(native):
    """C5, Matern rows (n = 10^6): the preconditioned log det with 16 and with 48 probes (different
    seeds) agree within their standard errors, which are below 1e-5 of the value -- the plain
    Lanczos quadrature of 128 probes on the same operator (bench: nll_grad_to_stall) carries +-26
    after 3000 iterations; here 48 solves of < 100 iterations."""
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    D, Q, R, m0, N = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m0, kern='matern')
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    M = K.preconditioner
    assert M is not None and not M.exact and K.device_operator().factor_mode == 3
    e1, s1, it1 = M.logdet_estimate(n_probes=16)
    e2, s2, it2 = M.logdet_estimate(n_probes=48, seed=99)
    assert it1.max() < 100 and it2.max() < 100
    assert s1 < 1e-5 * abs(e1) and s2 < 1e-5 * abs(e2), (s1, s2, e1)
    assert abs(e1 - e2) <= 4.0 * np.hypot(s1, s2), (e1, e2, s1, s2)
