"""CPU run of the parity suite: the SAME kernel source compiled with g++ on
top of the thread-level emulator (tests/emu).  Checks kernel index
arithmetic, barriers and the whole host stack without a GPU; the GPU run is
tests/test_gpu_suite.py.  The emulator is test infrastructure only."""
import pytest

import parity_suite as ps
from cases import DENSE_CASES


@pytest.fixture(scope='module', autouse=True)
def emu_library():
    from runlmc_amd import _lib, build
    lib = _lib.use_library(build.build_emu())
    assert not lib.is_hip
    yield lib
    _lib.use_library(None)


def test_bttb_examples():
    ps.check_bttb_examples()


def test_toeplitz_examples():
    ps.check_toeplitz_examples()


def test_operator_errors():
    ps.check_operator_errors()


def test_failed_setter_leaves_operator():
    ps.check_failed_setter_leaves_operator()


def test_kronecker_and_sum():
    ps.check_kronecker_and_sum()


def test_small_algebra():
    ps.check_small_algebra()


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_c1', 'lmc_q1', 'lmc_mid',
                                  'lmc_2d', 'fx2007', 'weather'])
def test_lmc_operator(name):
    ps.check_lmc_operator(name)


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_q1', 'lmc_2d'])
def test_solver_minres(name):
    ps.check_solver(name, minres=True)


def test_solver_cg():
    ps.check_solver('lmc_small', minres=False)


def test_solver_reference_rule():
    print(ps.check_solver_reference_rule())


def test_solver_edge_cases():
    ps.check_solver_edge_cases()


@pytest.mark.parametrize('name', DENSE_CASES + ['fx2007'])
def test_gradients_fixed_solves(name):
    ps.check_gradients_fixed_solves(name)


def test_gradients_end_to_end():
    ps.check_gradients_end_to_end('lmc_small')


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_q1'])
def test_logdet_slq(name):
    ps.check_logdet_slq(name)


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_2d'])
def test_model_prediction(name):
    ps.check_model_prediction(name)


def test_model_optimize():
    ps.check_model_optimize('lmc_q1')


@pytest.mark.parametrize('D,Q,m,nvec', [
    (2, 2, 1100, 3),      # 4096 = 64 x 64 (fused power-of-two kernels)
    (3, 2, 700, 3),       # 1536 = 3 * 512
    (2, 2, 2400, 3),      # 5120 = 5 * 1024
    (2, 1, 4500, 2),      # 9216 = 9 * 1024
    (2, 1, 7500, 2),      # 15360 = 15 * 1024
    (2, 2, 6300, 3),      # 12800 = 25 * 512
    (4, 3, 5004, 3)])     # C2: 10240 = 5 * 2048
def test_embedding_lengths(D, Q, m, nvec):
    """Every family of embedding length against the oracle (which always uses
    the reference's next power of two: the Toeplitz product does not depend
    on the embedding length)."""
    import numpy as np
    from oracle import operators as ops
    from runlmc_amd._native import GridOp
    rng = np.random.RandomState(m)
    g = GridOp(D, m, Q)
    tops = np.array([np.exp(-(0.002 + 0.01 * q) * np.arange(m)) for q in range(Q)])
    A = [rng.randn(1 + q % 2, D) for q in range(Q)]
    kap = [np.abs(rng.randn(D)) for _ in range(Q)]
    g.set_lmc(tops, A, kap)
    X = rng.randn(nvec, D * m)
    Y = g.matmat_host(X)
    Bs = ops.coreg_mats(A, kap)
    toeps = [ops.BTTBOracle(t) for t in tops]
    for v in range(nvec):
        ref = ops.grid_sum_matvec(Bs, toeps, X[v])
        assert np.abs(Y[v] - ref).max() < 1e-11 * np.abs(ref).max()


def test_unsorted_inputs():
    ps.check_unsorted_inputs()


def test_split_kernels():
    ps.check_split_kernels()


def test_ragged_and_empty_outputs():
    ps.check_ragged_and_empty_outputs()


def test_solver_fusions():
    ps.check_solver_fusions()


def test_solver_workspace_reuse():
    ps.check_solver_workspace_reuse()


def test_staged_wt_product():
    ps.check_staged_wt_product()


def test_w_poly_product():
    ps.check_w_poly_product()


def test_row_polynomial_form():
    ps.check_row_polynomial_form()


def test_generate_probe_dtypes():
    ps.check_generate_probe_dtypes()


def test_chunked_product():
    ps.check_chunked_product()


def test_single_tile_product():
    ps.check_single_tile_product()


def test_many_outputs():
    ps.check_many_outputs()


def test_rank_above_outputs():
    ps.check_rank_above_outputs()


def test_row_kernel_shapes():
    ps.check_row_kernel_shapes()


def test_cross_dots():
    ps.check_cross_dots()


def test_polynomial_bound():
    ps.check_polynomial_bound()


def test_polynomial_form():
    ps.check_polynomial_form()


def test_filter_form():
    ps.check_filter_form()


def test_minres_p_in_w():
    ps.check_minres_p_in_w()


def test_slfm_identity_quirk():
    ps.check_slfm_identity_quirk()


def test_polynomial_gate_boundary():
    ps.check_polynomial_gate_boundary()


def test_polynomial_rounds():
    ps.check_polynomial_rounds()


# --- round 6: direct solves through the polynomial form ---------------------------------
@pytest.mark.parametrize('kern,m_data', [('rbf', 400), ('periodic', 600)])
def test_direct_solve(kern, m_data):
    print(ps.check_direct_solve(kern, m_data))


def test_direct_unavailable():
    ps.check_direct_unavailable()


def test_precond_hi():
    ps.check_precond_hi()


def test_precond_hi_three_blocks(monkeypatch):
    """144 functions: the one-pass matrix-core expansion (k_hz_expand_mm, from three blocks on)."""
    monkeypatch.setenv('RUNLMC_PRECOND_HI_RANK', '144')
    out = ps.check_precond_hi(m_data=1400)
    assert out['hi', 3] <= 8, out


def test_logdet_preconditioned():
    ps.check_logdet_preconditioned()


def test_precond_hi_mixed_rows():
    out = ps.check_precond_hi(kern='mix', Q=3)
    assert out['forms'] == [1, 1, 2] or sorted(set(out['forms'])) == [1, 2], out


def test_direct_golden():
    print(ps.check_direct_golden())


def test_direct_row_orders():
    ps.check_direct_row_orders()


def test_small_batch_polynomial():
    print(ps.check_small_batch_polynomial())


def test_round6_abi_errors():
    ps.check_round6_abi_errors()


def test_device_probes():
    ps.check_device_probes()
