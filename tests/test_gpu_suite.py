"""GPU run of the parity suite: real HIP library on an MI355X, through the C
ABI.  pytest -m gpu."""
import pytest

import parity_suite as ps
from cases import DENSE_CASES, ALL_CASES, DATASET_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def hip_library():
    from runlmc_amd import _lib
    _lib.use_library(None)
    lib = _lib.get_library()
    assert lib.is_hip, 'GPU tests must run against librunlmc_hip.so'
    return lib


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


@pytest.mark.parametrize('name', ALL_CASES + DATASET_CASES)
def test_lmc_operator(name):
    ps.check_lmc_operator(name)


@pytest.mark.parametrize('name', ALL_CASES + DATASET_CASES)
def test_solver_minres(name):
    ps.check_solver(name, minres=True)


@pytest.mark.parametrize('name', DENSE_CASES)
def test_solver_cg(name):
    ps.check_solver(name, minres=False)


def test_solver_reference_rule():
    print(ps.check_solver_reference_rule())


def test_solver_edge_cases():
    ps.check_solver_edge_cases()


@pytest.mark.parametrize('name', DENSE_CASES + ['fx2007'])
def test_gradients_fixed_solves(name):
    ps.check_gradients_fixed_solves(name)


@pytest.mark.parametrize('name', DENSE_CASES + ['fx2007'])
def test_gradients_end_to_end(name):
    ps.check_gradients_end_to_end(name)


@pytest.mark.parametrize('name', DENSE_CASES + ['fx2007'])
def test_logdet_slq(name):
    ps.check_logdet_slq(name)


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_2d'])
def test_model_prediction(name):
    ps.check_model_prediction(name)


def test_model_optimize():
    ps.check_model_optimize('lmc_q1')


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


def test_block_cg_weather():
    ps.check_block_cg_weather()


def test_single_tile_product():
    ps.check_single_tile_product()


def test_many_outputs():
    ps.check_many_outputs()


def test_rank_above_outputs():
    ps.check_rank_above_outputs()


def test_row_kernel_shapes():
    ps.check_row_kernel_shapes([(4, 3, 5004, 5), (3, 2, 20001, 3), (2, 2, 30000, 3),
                                (2, 1, 70000, 2), (16, 2, 66000, 3), (13, 2, 40000, 2),
                                (12, 3, 26000, 4), (9, 1, 12000, 5), (1, 1, 9000, 7)])


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
@pytest.mark.parametrize('kern,m_data', [('rbf', 400), ('periodic', 600), ('rbf', 1500)])
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


def test_many_rhs_row_polynomial():
    ps.check_many_rhs_row_polynomial()


def test_round6_abi_errors():
    ps.check_round6_abi_errors()


def test_device_probes():
    ps.check_device_probes()
