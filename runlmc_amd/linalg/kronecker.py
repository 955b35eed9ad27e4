"""Kronecker product A (x) B of two operators (mirror of reference
runlmc/linalg/kronecker.py:12-52).

Generic operands go through the reference's two-step reshape algorithm on
the host (each step a batched matmat of the operand).  The LMC case -- a
dense coregionalisation matrix times a symmetric Toeplitz matrix -- is one
device operator (rl_gridop_set_dense + rl_gridop_mvm)."""
import numpy as np

from .matrix import Matrix, check_vector, check_block
from .numpy_matrix import NumpyMatrix
from .bttb import BTTB
from .._native import GridOp


def _is_lmc_term(K):
    return (isinstance(K, Kronecker) and isinstance(K.A, NumpyMatrix)
            and isinstance(K.B, BTTB) and K.A.is_square()
            and np.allclose(K.A.A, K.A.A.T, rtol=1e-13, atol=0))


def fuse_kronecker_sum(terms):
    """One GridOp for sum_q B_q (x) T_q, or None if `terms` is not of that
    form (symmetric dense B_q of one size, 1-D BTTB of one size; any D -- above 16
    outputs the handle is the 'wide' operator of csrc/rl_gridop.hip)."""
    if not all(_is_lmc_term(K) for K in terms):
        return None
    D = terms[0].A.shape[0]
    m = terms[0].B.shape[0]
    if any(K.A.shape[0] != D or K.B.shape[0] != m for K in terms):
        return None
    op = GridOp(D, m, len(terms))
    op.set_dense(np.stack([K.B.top for K in terms]),
                 np.stack([K.A.A for K in terms]))
    return op


class Kronecker(Matrix):
    def __init__(self, A, B):
        super().__init__(A.shape[0] * B.shape[0], A.shape[1] * B.shape[1])
        self.A = A
        self.B = B
        self._fused = None
        self._fused_tried = False

    def _try_fuse(self):
        if not self._fused_tried:
            self._fused_tried = True
            self._fused = fuse_kronecker_sum([self])
        return self._fused

    def as_numpy(self):
        return np.kron(self.A.as_numpy(), self.B.as_numpy())

    def _generic(self, x):
        # row-major vec trick: (A (x) B) vec(X) = vec(A X B^T), done as two
        # batched operator applications
        for M in (self.B, self.A):
            x = M.matmat(x.reshape(-1, M.shape[1]).T)
        return x.reshape(-1)

    def matvec(self, x):
        x = check_vector(x, self.shape[1])
        op = self._try_fuse()
        if op is not None:
            return op.matmat_host(x.astype(np.float64))
        return self._generic(np.asarray(x, dtype=np.float64))

    def matmat(self, X):
        X = check_block(X, self.shape[1])
        op = self._try_fuse()
        if op is not None:
            return op.matmat_host(np.ascontiguousarray(X.T, dtype=np.float64)).T
        return super().matmat(X)

    def upper_eig_bound(self):
        return self.A.upper_eig_bound() * self.B.upper_eig_bound()

    def __getstate__(self):
        state = super().__getstate__()
        state['_fused'] = None
        state['_fused_tried'] = False
        return state

    def __str__(self):
        return 'Kron(A, B)\nA\n{!s}\nB\n{!s}'.format(self.A, self.B)
