"""Diagonal operator (mirror of reference runlmc/linalg/diag.py:9-39)."""
import numpy as np

from .matrix import Matrix


class Diag(Matrix):
    def __init__(self, v):
        v = np.asarray(v)
        if v.ndim != 1:
            raise ValueError('Expected input vector for Diagonal matrix, '
                             'got something of shape {}'.format(v.shape))
        super().__init__(len(v), len(v))
        self.v = v

    def matvec(self, x):
        return x * self.v

    def matmat(self, X):
        return self.v[:, None] * X

    def as_numpy(self):
        return np.diag(self.v)

    def upper_eig_bound(self):
        return self.v.max()

    def __str__(self):
        return 'Diag(len {}): {}'.format(len(self.v), self.v)
