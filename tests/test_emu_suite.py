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


def test_kronecker_and_sum():
    ps.check_kronecker_and_sum()


def test_small_algebra():
    ps.check_small_algebra()


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_c1', 'lmc_q1', 'lmc_mid',
                                  'fx2007', 'weather'])
def test_lmc_operator(name):
    ps.check_lmc_operator(name)


@pytest.mark.parametrize('name', ['lmc_small', 'lmc_q1'])
def test_solver_minres(name):
    ps.check_solver(name, minres=True)


def test_solver_cg():
    ps.check_solver('lmc_small', minres=False)


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


def test_model_prediction():
    ps.check_model_prediction('lmc_small')


def test_model_optimize():
    ps.check_model_optimize('lmc_q1')
