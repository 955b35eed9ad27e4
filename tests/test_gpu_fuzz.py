"""Randomised shapes through the grid operator on the GPU: every combination
of outputs, kernels, ranks, grid length and batch size picks its own embedding
length, split, tile sizes and kernel family; each result is held to the oracle.
RUNLMC_FUZZ_N raises the number of draws (default 40)."""
import os

import numpy as np
import pytest

from oracle import operators as ops

pytestmark = pytest.mark.gpu


def test_random_shapes_vs_oracle():
    from runlmc_amd._native import GridOp
    n_draws = int(os.environ.get('RUNLMC_FUZZ_N', '40'))
    rng = np.random.RandomState(20260101)
    worst = 0.0
    for draw in range(n_draws):
        D = int(rng.randint(1, 17))
        Q = int(rng.randint(1, 5))
        m = int(np.exp(rng.uniform(0, np.log(6000))))
        nvec = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 70, 130]))
        if D * m * nvec > 3e6:
            nvec = max(1, int(3e6 // (D * m)))
        sizes = None
        if rng.rand() < 0.25:                       # a two-dimensional (BTTB) grid
            sizes = (int(rng.randint(1, 60)), int(rng.randint(1, 60)))
            m = sizes[0] * sizes[1]
            if D * m * nvec > 3e6:
                nvec = max(1, int(3e6 // (D * m)))
            i1, i2 = np.meshgrid(np.arange(sizes[0]), np.arange(sizes[1]), indexing='ij')
            r = np.sqrt(i1 ** 2 + (0.7 * i2) ** 2).ravel()
            tops = np.array([np.exp(-(0.05 + 0.3 * rng.rand()) * r) for _ in range(Q)])
        else:
            tops = np.array([np.exp(-(0.01 + 0.2 * rng.rand()) *
                                    np.arange(m) ** (1 + 0.4 * rng.rand())) for _ in range(Q)])
        A = [rng.randn(int(rng.randint(0, 3)), D) for _ in range(Q)]
        A = [a if len(a) else None for a in A]
        kap = [np.abs(rng.randn(D)) + 0.05 for _ in range(Q)]
        g = GridOp(D, m, Q, sizes=sizes)
        g.set_lmc(tops, A, kap)
        X = rng.randn(nvec, D * m)
        Y = g.matmat_host(X)
        Bs = ops.coreg_mats([a if a is not None else np.zeros((0, D)) for a in A], kap)
        toeps = [ops.BTTBOracle(t, sizes) if sizes else ops.BTTBOracle(t) for t in tops]
        for v in sorted(set([0, nvec - 1, nvec // 2])):
            ref = ops.grid_sum_matvec(Bs, toeps, X[v])
            err = np.abs(Y[v] - ref).max() / max(np.abs(ref).max(), 1e-300)
            worst = max(worst, err)
            assert err < 1e-11, (draw, D, Q, m, sizes, nvec, v, g.L, g.N1, g.N2, err)
    print('fuzz: %d shapes, worst relative error %.2e' % (n_draws, worst))


def test_random_structured_forms_vs_oracle():
    """Randomised kernels through the per-top FORMS of the product with the batch gate
    lifted: RBF / periodic rows (polynomial form at whatever rank the verification
    accepts: 24 ... 48, short grids from 96 points on), Matern-3/2 / its derivative /
    exponential rows (recursive filter), a kinked row now and then (transform kernels for
    the operator), 1 ... 28 outputs (above 16: the wide operator).  Every product against
    the oracle at 1e-11 and against the transform kernels of the same handle at 1e-12."""
    from runlmc_amd._native import GridOp
    n_draws = int(os.environ.get('RUNLMC_FUZZ_N', '40'))
    rng = np.random.RandomState(20261003)
    worst, seen = 0.0, {}
    for draw in range(n_draws):
        D = int(rng.choice([1, 2, 3, 4, 7, 8, 10, 13, 16, 17, 19, 28]))
        Q = int(rng.randint(1, 5))
        m = int(np.exp(rng.uniform(np.log(96), np.log(9000))))
        nvec = int(rng.choice([1, 2, 3, 9, 17, 40]))
        if D * m * nvec > 2e6:
            nvec = max(1, int(2e6 // (D * m)))
        x = np.linspace(0, 1 + 2 * rng.rand(), m)
        tops = []
        for q in range(Q):
            kind = rng.choice(['rbf', 'rbf', 'periodic', 'matern', 'dmatern', 'exp', 'kink'],
                              p=[0.25, 0.15, 0.15, 0.2, 0.1, 0.1, 0.05])
            gam = float(np.exp(rng.uniform(np.log(0.5), np.log(60.0))))
            if kind == 'rbf':
                tops.append(np.exp(-0.5 * gam * x ** 2))
            elif kind == 'periodic':
                tops.append(np.exp(-0.5 * np.sin(np.pi * x / (0.8 + 4 * rng.rand())) ** 2))
            elif kind == 'matern':
                tops.append((1 + np.sqrt(3) * gam * x) * np.exp(-np.sqrt(3) * gam * x))
            elif kind == 'dmatern':
                tops.append(-3.0 * gam * x * x * np.exp(-np.sqrt(3) * gam * x))
            elif kind == 'exp':
                tops.append(np.exp(-gam * x))
            else:
                tops.append(1.0 / (1.0 + gam * x))
        tops = np.array(tops)
        A = [rng.randn(int(rng.randint(0, 3)), D) for _ in range(Q)]
        A = [a if len(a) else None for a in A]
        kap = [np.abs(rng.randn(D)) + 0.05 for _ in range(Q)]
        g = GridOp(D, m, Q)
        g.set_lmc(tops, A, kap)
        forms, structured = g.top_forms()
        X = rng.randn(nvec, D * m)
        g.set_form_gate(0)
        Y = g.matmat_host(X)
        Yt = [g.matmat_host(X[:1], top=q)[0] for q in range(Q)]
        g.set_form_gate(1 << 62)
        F = g.matmat_host(X)
        g.set_form_gate(-1)
        Bs = ops.coreg_mats([a if a is not None else np.zeros((0, D)) for a in A], kap)
        toeps = [ops.BTTBOracle(t) for t in tops]
        key = (tuple(forms), bool(structured), D > 16)
        seen[key] = seen.get(key, 0) + 1
        for v in sorted(set([0, nvec - 1])):
            ref = ops.grid_sum_matvec(Bs, toeps, X[v])
            err = np.abs(Y[v] - ref).max() / max(np.abs(ref).max(), 1e-300)
            worst = max(worst, err)
            assert err < 1e-11, (draw, D, Q, m, nvec, forms, structured, err)
        scale = np.maximum(np.abs(F).max(axis=1, keepdims=True), 1e-300)
        assert (np.abs(Y - F) / scale).max() < 1e-12, (draw, D, Q, m, forms)
        for q in range(Q):
            ref = np.concatenate([toeps[q].matvec(r) for r in X[0].reshape(D, m)])
            err = np.abs(Yt[q] - ref).max() / max(np.abs(ref).max(), 1e-300)
            assert err < 1e-11, (draw, 'top', q, D, m, forms, err)
    print('structured-form fuzz: %d shapes, worst relative error %.2e, %d distinct '
          '(forms, structured, wide) combinations' % (n_draws, worst, len(seen)))
    assert len(seen) >= 6


@pytest.mark.parametrize('staged', [False, True])
def test_random_ski_operators_and_solves(staged, monkeypatch):
    """Random ragged multi-output problems through the package API: the SKI
    operator against the oracle's, then a batched solve whose reported
    residuals are recomputed through the oracle operator.  `staged` forces the
    LDS-staged W^T / W products of large batches onto these small ones, with
    the fused small-system forms off so that the solver goes through them."""
    if staged:
        monkeypatch.setenv('RUNLMC_STAGED_WT', '1')
        monkeypatch.setenv('RUNLMC_NO_FUSE_W', '1')
        monkeypatch.setenv('RUNLMC_NO_FUSE_WT', '1')
    from runlmc_amd.approx.interpolation import autogrid, multi_interpolant
    from runlmc_amd.approx.iterative import Iterative
    from runlmc_amd.kern.stationary import RBF, Matern32
    from runlmc_amd.lmc.functional_kernel import FunctionalKernel
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from oracle import likelihood as olik
    from oracle.kernels import KernelSpec, RBFSpec, Matern32Spec
    n_draws = int(os.environ.get('RUNLMC_FUZZ_N', '40')) // 2
    rng = np.random.RandomState(77)
    for draw in range(n_draws):
        D = int(rng.randint(1, 7))
        Q = int(rng.randint(1, 4))
        lens = [int(rng.choice([0, 3, 17, 60, 250, 700])) for _ in range(D)]
        if sum(lens) < 8:
            lens[0] = 40
        Xs = [rng.rand(n, 1) * (1 + 3 * rng.rand()) for n in lens]
        m = int(rng.choice([12, 40, 150, 600]))
        grid = autogrid([X for X in Xs if len(X)], None, None, np.array([float(m)]))[0]
        W = multi_interpolant(Xs, grid)
        WT = W.transpose().tocsr()
        kinds = [int(rng.randint(0, 2)) for _ in range(Q)]
        scales = [float(np.exp(rng.uniform(0, 4))) for _ in range(Q)]
        ranks = [int(rng.randint(1, 3)) for _ in range(Q)]
        fk = FunctionalKernel(D=D, lmc_kernels=[RBF(s) if k == 0 else Matern32(s)
                                                for k, s in zip(kinds, scales)], lmc_ranks=ranks)
        fk.noise = 0.05 + rng.rand(D)
        fk.set_input_dim(1)
        ad = (0,)
        dists = grid - grid[0]
        K, _ = gen_grid_kernel(fk, {ad: dists}, {ad: (W, WT)}, lens)
        spec = KernelSpec(D, [RBFSpec(s) if k == 0 else Matern32Spec(s)
                              for k, s in zip(kinds, scales)], fk.coreg_vecs, fk.coreg_diags,
                          fk.noise)
        spec.set_input_dim(1)
        op = olik.LMCOperatorOracle(spec, dists, W, WT, lens)
        n = sum(lens)
        k = int(rng.choice([1, 2, 5]))
        X = rng.randn(k, n)
        got = K.matmat(X.T).T
        ref = np.array([op.matvec(v) for v in X])
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < 1e-11, (draw, D, Q, lens, m, err)
        sol, iters, resid = Iterative.solve(K, X, verbose=True, tol=1e-6)
        for i in range(k):
            true_res = np.linalg.norm(X[i] - op.matvec(sol[i]))
            assert abs(true_res - resid[i]) <= 1e-9 + 1e-6 * true_res, (draw, i, true_res, resid[i])
