"""Operator protocol (mirror of reference runlmc/linalg/matrix.py:7-90).

Callers rely on: ``shape``, ``dtype == float64``, ``matvec(x)`` returning a
NEW 1-D array, ``matmat(X)`` for an (n, k) block of columns, ``as_numpy()``,
``as_linear_operator()`` (cached scipy LinearOperator), ``Matrix.wrap``, and
picklability.  Device-backed subclasses override ``matmat`` so that all k
columns go through one batched kernel launch instead of the reference's
column loop (matrix.py:55-67).
"""
import numpy as np
import scipy.sparse.linalg


class Matrix:
    def __init__(self, n, m):
        if n < 1 or m < 1:
            raise ValueError('Size of the matrix {} < 1'.format((n, m)))
        self.dtype = np.float64
        self.shape = (n, m)
        self._op = None

    def as_linear_operator(self):
        if self._op is None:
            self._op = scipy.sparse.linalg.LinearOperator(
                shape=self.shape, dtype=self.dtype,
                matvec=self.matvec, matmat=self.matmat)
        return self._op

    def as_numpy(self):
        return self.matmat(np.identity(self.shape[1]))

    def matvec(self, x):
        raise NotImplementedError

    def matmat(self, X):
        X = np.asarray(X)
        cols = [self.matvec(X[:, j]) for j in range(X.shape[1])]
        return np.stack(cols, axis=1) if cols else np.empty((self.shape[0], 0))

    def is_square(self):
        return self.shape[0] == self.shape[1]

    @staticmethod
    def wrap(shape, mvm):
        return _Wrapped(shape, mvm)

    # the cached LinearOperator holds bound methods; drop it when pickling
    def __getstate__(self):
        state = dict(self.__dict__)
        state['_op'] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)


class _Wrapped(Matrix):
    def __init__(self, shape, mvm):
        super().__init__(*shape)
        self._mvm = mvm

    def matvec(self, x):
        return self._mvm(x)


def check_vector(x, n, what='x'):
    x = np.asarray(x)
    if x.ndim != 1 or x.shape[0] != n:
        raise ValueError('{} must be a vector of length {}, got shape {}'
                         .format(what, n, x.shape))
    return x


def check_block(X, n, what='X'):
    X = np.asarray(X)
    if X.ndim != 2 or X.shape[0] != n:
        raise ValueError('{} must have shape ({}, k), got {}'
                         .format(what, n, X.shape))
    return X
