"""Pin the oracle: every oracle function against the vectors the REAL
reference produced (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import operators as ops
from oracle import interp
from oracle import likelihood as lik
from oracle.solver import iterative_solve
from cases import Case, ALL_CASES, DENSE_CASES, DATASET_CASES, SMOOTH_CASES, GOLDEN

import os

# the reference's own operator tests use rtol = atol = 1e-6
# (runlmc/linalg/test_matrix_base.py:33-47); the oracle restates the same
# NumPy calls, so it is held to roundoff instead.
TIGHT = dict(rtol=1e-12, atol=1e-12)


@pytest.fixture(scope='module')
def lin():
    return np.load(os.path.join(GOLDEN, 'linalg.npz'))


def test_bttb_examples(lin):
    for i in range(int(lin['bttb_count'])):
        top, sizes = lin[f'bttb{i}_top'], lin[f'bttb{i}_sizes']
        M = ops.BTTBOracle(top, sizes)
        n = top.size
        np.testing.assert_array_equal(M.as_numpy(), lin[f'bttb{i}_dense'])
        np.testing.assert_allclose(M.matvec(np.arange(n) + 1),
                                   lin[f'bttb{i}_matvec'], **TIGHT)
        np.testing.assert_allclose(
            M.matmat(np.arange(2 * n).reshape(-1, 2)),
            lin[f'bttb{i}_matmat'], **TIGHT)
        # and the reference's own assertion: matvec == dense @ x at 1e-6
        np.testing.assert_allclose(M.matvec(np.arange(n) + 1),
                                   M.as_numpy().dot(np.arange(n) + 1),
                                   rtol=1e-6, atol=1e-6)


def test_bttb_known_layouts(lin):
    for tag in ('2d', '3d'):
        top = lin[f'bttb_known{tag}_top']
        np.testing.assert_array_equal(
            ops.bttb_dense(top.ravel(), top.shape),
            lin[f'bttb_known{tag}_dense'])


def test_bttb_errors():
    with pytest.raises(ValueError):
        ops.BTTBOracle(np.arange(8).reshape(2, 4), (2, 4))
    with pytest.raises(ValueError):
        ops.BTTBOracle(np.array([]), ())
    with pytest.raises(ValueError):
        ops.BTTBOracle(np.arange(8.), (3, 4))
    with pytest.raises(TypeError):
        ops.BTTBOracle(np.arange(5) * 1j, (5,))


def test_toeplitz_examples(lin):
    for i in range(int(lin['toep_count'])):
        top = lin[f'toep{i}_top']
        n = len(top)
        np.testing.assert_allclose(ops.toeplitz_matvec(top, np.arange(n) + 1),
                                   lin[f'toep{i}_matvec'], **TIGHT)
        X = np.arange(2 * n).reshape(-1, 2)
        got = np.stack([ops.toeplitz_matvec(top, c) for c in X.T], axis=1)
        np.testing.assert_allclose(got, lin[f'toep{i}_matmat'], **TIGHT)
        # pow2 embedding gives the same Toeplitz product
        np.testing.assert_allclose(
            ops.BTTBOracle(top).matvec(np.arange(n) + 1),
            lin[f'toep{i}_dense'].dot(np.arange(n) + 1), rtol=1e-9, atol=1e-9)


def test_kronecker_and_sum(lin):
    for i in range(int(lin['kron_count'])):
        B, top = lin[f'kron{i}_B'], lin[f'kron{i}_top']
        T = ops.BTTBOracle(top)
        n = B.shape[0] * len(top)
        np.testing.assert_allclose(ops.kron_matvec(B, T, np.arange(n) + 1),
                                   lin[f'kron{i}_matvec'], rtol=1e-12,
                                   atol=1e-9)
        np.testing.assert_allclose(
            np.kron(B, T.as_numpy()), lin[f'kron{i}_dense'], **TIGHT)
    Bs, tops, x = lin['sum_Bs'], lin['sum_tops'], lin['sum_x']
    toeps = [ops.BTTBOracle(t) for t in tops]
    np.testing.assert_allclose(ops.grid_sum_matvec(Bs, toeps, x),
                               lin['sum_matvec'], rtol=1e-12, atol=1e-10)


def test_interpolation():
    g = np.load(os.path.join(GOLDEN, 'interp.npz'))
    np.testing.assert_allclose(interp.keys_cubic(g['cubic_in']),
                               g['cubic_out'], **TIGHT)
    np.testing.assert_allclose(
        interp.cubic_rows(g['ic_grid'], g['ic_samples']), g['ic_dense'],
        **TIGHT)
    Xs = [g['mi_X0'], g['mi_X1'], g['mi_X2']]
    grid = interp.auto_grid_1d(Xs)
    np.testing.assert_allclose(grid, g['ag_default'], **TIGHT)
    np.testing.assert_allclose(interp.auto_grid_1d(Xs, m=25), g['ag_m25'],
                               **TIGHT)
    np.testing.assert_allclose(interp.multi_interp(Xs, grid).toarray(),
                               g['mi_dense'], **TIGHT)


@pytest.mark.parametrize('name', ALL_CASES + DATASET_CASES + SMOOTH_CASES)
def test_lmc_operator(name):
    c = Case(name)
    spec = c.spec()
    np.testing.assert_allclose(
        np.reshape(spec.eval_kernels_fixed_dim(c.grid_dists, c.ad), (c.Q, -1)),
        c.g['tops'], **TIGHT)
    for q, gl in enumerate(spec.eval_kernel_gradients({c.ad: c.grid_dists})):
        for p, gq in enumerate(gl):
            np.testing.assert_allclose(np.ravel(gq), c.g[f'dtop{q}_{p}'], **TIGHT)
    for kt in ('sum', 'bt', 'slfm'):
        op = lik.LMCOperatorOracle(spec, c.grid_dists, c.W, c.WT, c.lens,
                                   ktype=kt, active_dim=c.ad)
        got = np.array([op.grid_matvec(v) for v in c.g['grid_x']])
        ref = c.g[f'grid_mv_{kt}']
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * scale)
    op = lik.LMCOperatorOracle(spec, c.grid_dists, c.W, c.WT, c.lens, active_dim=c.ad)
    got = np.array([op.matvec(v) for v in c.g['full_x']])
    ref = c.g['full_mv']
    np.testing.assert_allclose(got, ref, rtol=0,
                               atol=1e-12 * np.abs(ref).max())


def test_dense_and_logdet():
    c = Case('lmc_small')
    op = lik.LMCOperatorOracle(c.spec(), c.grid_dists, c.W, c.WT, c.lens)
    Kd = op.as_numpy()
    np.testing.assert_allclose(0.5 * (Kd + Kd.T), c.g['K_dense'], rtol=0,
                               atol=1e-12)
    np.testing.assert_allclose(lik.logdet_dense(0.5 * (Kd + Kd.T)),
                               float(c.g['logdet_dense']), rtol=1e-12)


@pytest.mark.parametrize('name', SMOOTH_CASES)
def test_dense_solves_and_logdet_smooth(name):
    """The oracle's dense K~ of the smooth case: alpha, K~^-1 r_i and log det K~ against the
    reference's own dense Cholesky values (what the device's direct solve is held to)."""
    import scipy.linalg as la
    c = Case(name)
    op = lik.LMCOperatorOracle(c.spec(), c.grid_dists, c.W, c.WT, c.lens)
    Kd = op.as_numpy()
    Kd = 0.5 * (Kd + Kd.T)
    cf = la.cho_factor(Kd)
    np.testing.assert_allclose(2 * np.log(np.diag(cf[0])).sum(), float(c.g['logdet_dense']),
                               rtol=1e-11)
    ref = c.g['alpha_dense']
    np.testing.assert_allclose(la.cho_solve(cf, c.y), ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    ref = c.g['inv_rs_dense']
    np.testing.assert_allclose(la.cho_solve(cf, c.rs.T.astype(float)).T, ref, rtol=0,
                               atol=1e-8 * np.abs(ref).max())


def test_exact_dense_twin():
    """oracle.likelihood.exact_gradients against the reference's own ExactLMCLikelihood
    (likelihood.py:137-217, exact_deriv.py) run on a small seeded model (exact_small.npz,
    make_golden.py: gen_exact): the dense twin the reference's `bench.py opt` measures its
    err:grad and alpha error columns against (benchmarks/benchlib/bench.py:235-283)."""
    from oracle.kernels import KernelSpec, RBFSpec, Matern32Spec, StdPeriodicSpec
    g = np.load(os.path.join(GOLDEN, 'exact_small.npz'), allow_pickle=True)
    D, Q = int(g['D']), int(g['Q'])
    make = {'rbf': RBFSpec, 'matern': Matern32Spec, 'periodic': StdPeriodicSpec}
    kerns = []
    for d in g['kdesc']:
        parts = str(d).split(';')
        kerns.append(make[parts[0]](*[float(v) for v in parts[1:]]))
    spec = KernelSpec(D, kerns, [g[f'A{q}'] for q in range(Q)],
                      [g[f'kappa{q}'] for q in range(Q)], g['noise'])
    spec.set_input_dim(1)
    Xs = [g[f'X{d}'] for d in range(D)]
    got, alpha, K = lik.exact_gradients(spec, Xs, g['y'])
    np.testing.assert_allclose(K, g['K'], rtol=0, atol=1e-13 * np.abs(g['K']).max())
    np.testing.assert_allclose(alpha, g['alpha'], rtol=0, atol=1e-9 * np.abs(g['alpha']).max())
    for q in range(Q):
        for mine, key in ((got['coreg_vec'][q], f'grad_A{q}'),
                          (got['coreg_diag'][q], f'grad_kappa{q}'),
                          (np.array(got['kernel'][q]), f'grad_kern{q}')):
            ref = g[key]
            np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-9 * max(1, np.abs(ref).max()))
    np.testing.assert_allclose(got['noise'], g['grad_noise'], rtol=0,
                               atol=1e-9 * np.abs(g['grad_noise']).max())


@pytest.mark.parametrize('name', DENSE_CASES + ['fx2007'] + SMOOTH_CASES)
def test_gradients_fixed_probes(name):
    """Reference gradient loops fed dense solves + stored probes: fully
    deterministic, so the oracle must match to roundoff."""
    c = Case(name)
    g = lik.stochastic_gradients(c.spec(), c.grid_dists, c.W, c.WT, c.lens,
                                 c.g['alpha_dense'], c.rs,
                                 c.g['inv_rs_dense'], active_dim=c.ad)
    for q in range(c.Q):
        for mine, key in ((g['coreg_vec'][q], f'grad_A{q}'),
                          (g['coreg_diag'][q], f'grad_kappa{q}'),
                          (np.array(g['kernel'][q]), f'grad_kern{q}')):
            ref = c.g[key]
            np.testing.assert_allclose(mine, ref, rtol=1e-9,
                                       atol=1e-9 * max(1, np.abs(ref).max()))
    np.testing.assert_allclose(g['noise'], c.g['grad_noise'], rtol=1e-9,
                               atol=1e-9)


@pytest.mark.parametrize('name', ALL_CASES + SMOOTH_CASES)
def test_reference_solver_wrapper(name):
    """The reference's Iterative.solve iterates, iteration counts and
    residuals (SciPy under it) vs the oracle's restated MINRES/CG."""
    c = Case(name)
    op = lik.LMCOperatorOracle(c.spec(), c.grid_dists, c.W, c.WT, c.lens,
                               active_dim=c.ad)
    rhs = [c.y] + [r.astype(float) for r in c.rs[:2]]
    for i, b in enumerate(rhs):
        x, it, err, _ = iterative_solve(op.matvec, b, tol=1e-4, minres=True)
        # the stopping test sits on roundoff of the operator (the reference's
        # slfm composition sums in another order): allow +-2 iterations
        assert abs(it - int(c.g['ref_minres_iters'][i])) <= 2
        ref = c.g['ref_minres_x'][i]
        # Krylov iterates amplify roundoff by cond(K); compare through the
        # residual and loosely through x
        assert abs(err - float(c.g['ref_minres_err'][i])) <= \
            1e-6 * max(1.0, err) + 0.05 * err
        np.testing.assert_allclose(x, ref, rtol=0,
                                   atol=1e-6 * np.abs(ref).max())
    x, it, err, _ = iterative_solve(op.matvec, c.y, tol=1e-4, minres=False)
    assert abs(it - int(c.g['ref_cg_iters'][0])) <= 2
    ref = c.g['ref_cg_x'][0]
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-6 * np.abs(ref).max())


def test_slfm_identity_quirk(golden_dir):
    """The oracle's statement of the reference's identity terms under 'slfm'
    (oracle.likelihood.slfm_identity_terms) against the reference's own GridKernel
    on a pure-SLFM and a pure-independent model (tests/golden/slfm_quirk.npz)."""
    import os
    import scipy.sparse as sp
    from oracle.kernels import KernelSpec, RBFSpec
    from oracle import likelihood as olik
    g = np.load(os.path.join(golden_dir, 'slfm_quirk.npz'))
    D, m = int(g['D']), int(g['m'])
    W = sp.identity(D * m, format='csr')
    for name in ('pure_slfm', 'pure_indep'):
        kd = [str(s_).split(';') for s_ in g[name + '_kdesc']]
        nl, ns = [int(v) for v in g[name + '_nums']]
        Q = len(kd)
        spec = KernelSpec(D, [RBFSpec(float(k[1])) for k in kd],
                          [g[f'{name}_A{q}'] for q in range(Q)],
                          [g[f'{name}_kappa{q}'] for q in range(Q)], 0.1 * np.ones(D),
                          num_lmc=nl, num_slfm=ns)
        spec.set_input_dim(1)
        assert olik.slfm_identity_terms(spec) == 1
        for quirk, key in ((False, '_sum'), (True, '_slfm')):
            op = olik.LMCOperatorOracle(spec, g['grid_dists'], W, W, [m] * D, ktype='slfm',
                                        reference_slfm_identity=quirk)
            got = np.array([op.grid_matvec(v) for v in g['x']])
            ref = g[name + key]
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
