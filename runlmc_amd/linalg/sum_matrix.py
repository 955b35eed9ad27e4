"""Sum of same-shaped operators (mirror of reference
runlmc/linalg/sum_matrix.py:8-45).

When every term is a Kronecker(NumpyMatrix, BTTB) over one grid, the whole
sum is ONE device operator (rl_gridop_mvm does sum_q B_q (x) T_q in a single
pass: 2D transforms per product instead of the reference's 2QD)."""
import numpy as np

from .matrix import Matrix, check_vector, check_block


class SumMatrix(Matrix):
    def __init__(self, Ks):
        if not Ks:
            raise ValueError('Need at least one matrix to sum')
        shapes = [K.shape for K in Ks]
        if len(set(shapes)) != 1:
            raise ValueError('At most one distinct shape expected in sum, '
                             'found shapes:\n{}'.format(shapes))
        super().__init__(*shapes[0])
        self.Ks = Ks
        self._fused = None
        self._fused_tried = False

    def _try_fuse(self):
        if not self._fused_tried:
            self._fused_tried = True
            from .kronecker import fuse_kronecker_sum
            self._fused = fuse_kronecker_sum(self.Ks)
        return self._fused

    def matvec(self, x):
        op = self._try_fuse()
        if op is not None:
            x = check_vector(x, self.shape[1])
            return op.matmat_host(x.astype(np.float64))
        total = self.Ks[0].matvec(x)
        for K in self.Ks[1:]:
            total = total + K.matvec(x)
        return total

    def matmat(self, X):
        op = self._try_fuse()
        if op is not None:
            X = check_block(X, self.shape[1])
            return op.matmat_host(np.ascontiguousarray(X.T, dtype=np.float64)).T
        total = self.Ks[0].matmat(X)
        for K in self.Ks[1:]:
            total = total + K.matmat(X)
        return total

    def as_numpy(self):
        total = self.Ks[0].as_numpy()
        for K in self.Ks[1:]:
            total = total + K.as_numpy()
        return total

    def upper_eig_bound(self):
        return sum(K.upper_eig_bound() for K in self.Ks)

    def __getstate__(self):
        state = super().__getstate__()
        state['_fused'] = None
        state['_fused_tried'] = False
        return state

    def __str__(self):
        return 'SumMatrix([..., Ki, ...])\n' + '\n'.join(
            'K{}\n{!s}'.format(i, K) for i, K in enumerate(self.Ks))
