"""Identity operator (mirror of reference runlmc/linalg/identity.py:10-25)."""
import numpy as np

from .matrix import Matrix


class Identity(Matrix):
    def __init__(self, n):
        super().__init__(n, n)

    def matvec(self, x):
        return x

    def matmat(self, X):
        return X

    def as_numpy(self):
        return np.identity(self.shape[0])

    def upper_eig_bound(self):
        return 1
