"""Product of operators, applied right to left (mirror of reference
runlmc/linalg/composition.py:8-22)."""
from .matrix import Matrix


class Composition(Matrix):
    def __init__(self, mats):
        super().__init__(mats[0].shape[0], mats[-1].shape[1])
        self.mats = mats

    def matvec(self, x):
        for M in self.mats[::-1]:
            x = M.matvec(x)
        return x

    def matmat(self, X):
        for M in self.mats[::-1]:
            X = M.matmat(X)
        return X
