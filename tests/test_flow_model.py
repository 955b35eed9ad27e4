"""The FFT flow graph the kernels implement vs numpy.fft and the oracle."""
import numpy as np
import pytest

import flow_model as fm
from oracle import operators as ops


@pytest.mark.parametrize('N,mr', [(8, 8), (16, 4), (32, 8), (64, 8), (128, 4)])
def test_dif_matches_fft(N, mr):
    rng = np.random.RandomState(N)
    x = rng.randn(N, 3) + 1j * rng.randn(N, 3)
    plan = fm.radix_plan(N, mr)
    a = x.copy()
    fm.dif_forward(a, plan)
    freq = fm.position_to_freq(N, plan)
    np.testing.assert_allclose(a, np.fft.fft(x, axis=0)[freq], atol=1e-11)
    fm.dif_adjoint(a, plan)
    np.testing.assert_allclose(a / N, x, atol=1e-12)


def test_four_step():
    N1, N2 = 16, 32
    rng = np.random.RandomState(0)
    z = rng.randn(N1 * N2) + 1j * rng.randn(N1 * N2)
    p1, p2 = fm.radix_plan(N1), fm.radix_plan(N2)
    S = fm.four_step_forward(z, N1, N2, p1, p2)
    k1, k2 = fm.position_to_freq(N1, p1), fm.position_to_freq(N2, p2)
    ref = np.fft.fft(z)
    np.testing.assert_allclose(S, ref[k1[:, None] + N1 * k2[None, :]],
                               atol=1e-10)
    back = fm.four_step_adjoint(S, N1, N2, p1, p2) / (N1 * N2)
    np.testing.assert_allclose(back, z, atol=1e-12)


def test_grid_mvm_model_vs_oracle():
    rng = np.random.RandomState(3)
    D, Q, m, nvec = 3, 2, 50, 3
    L = ops.next_pow2(2 * m)
    tops = [np.exp(-0.1 * (q + 1) * np.arange(m)) for q in range(Q)]
    A = [rng.randn(2, D), rng.randn(1, D)]
    kappa = np.abs(rng.randn(Q, D))
    Bs = ops.coreg_mats(A, list(kappa))
    facA = np.vstack(A)
    facW = np.ones(3)
    facQ = np.array([0, 0, 1])
    X = rng.randn(nvec, D, m)
    Y = fm.grid_mvm_model(tops, facA, facW, facQ, kappa, X, L, 8, L // 8)
    toeps = [ops.BTTBOracle(t) for t in tops]
    for v in range(nvec):
        ref = ops.grid_sum_matvec(Bs, toeps, X[v].ravel())
        np.testing.assert_allclose(Y[v].ravel(), ref, rtol=0,
                                   atol=1e-11 * np.abs(ref).max())
