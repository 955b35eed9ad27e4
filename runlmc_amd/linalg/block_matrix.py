"""Symmetric D x D array of equally sized square blocks (mirror of reference
runlmc/linalg/block_matrix.py:12-54)."""
import numpy as np
import scipy.linalg as la

from .matrix import Matrix


class SymmSquareBlockMatrix(Matrix):
    def __init__(self, blocks):
        self.D = len(blocks)
        if {len(row) for row in blocks} != {self.D}:
            raise ValueError('Uneven sizes')
        m = blocks[0][0].shape[0]
        super().__init__(self.D * m, self.D * m)
        self.blocks = blocks
        self._m = m

    def _slice(self, i):
        return slice(i * self._m, (i + 1) * self._m)

    def matvec(self, x):
        out = np.zeros(self.shape[0], dtype=self.dtype)
        for i in range(self.D):
            for j in range(self.D):
                out[self._slice(i)] += self.blocks[i][j].matvec(x[self._slice(j)])
        return out

    def as_numpy(self):
        out = np.zeros(self.shape)
        for i in range(self.D):
            for j in range(self.D):
                out[self._slice(i), self._slice(j)] = self.blocks[i][j].as_numpy()
        return out

    def upper_eig_bound(self):
        bounds = np.array([[self.blocks[min(i, j)][max(i, j)].upper_eig_bound()
                            for j in range(self.D)] for i in range(self.D)],
                          dtype=float)
        return la.norm(bounds, 1)

    def __str__(self):
        return 'SymmBlockMatrix(..., block(i,j), ...)\n' + '\n'.join(
            'block({},{})\n{!s}'.format(i, j, self.blocks[i][j])
            for i in range(self.D) for j in range(i, self.D))
