"""Dense adapter (mirror of reference runlmc/linalg/numpy_matrix.py:8-34)."""
import numpy as np

from .matrix import Matrix


class NumpyMatrix(Matrix):
    def __init__(self, nparr):
        nparr = np.asarray(nparr)
        if nparr.ndim != 2:
            raise ValueError('Input numpy array of shape {} not matrix'
                             .format(nparr.shape))
        self.A = nparr.astype('float64', casting='safe')
        super().__init__(*self.A.shape)

    def as_numpy(self):
        return self.A

    def matvec(self, x):
        return self.A.dot(x)

    def matmat(self, X):
        return self.A.dot(X)

    def upper_eig_bound(self):
        return np.abs(self.A).sum(axis=1).max()

    def __str__(self):
        return str(self.A)
