"""Symmetric Toeplitz operator (mirror of reference
runlmc/linalg/toeplitz.py:18-92).

The reference embeds in a circulant of length exactly 2n; the device kernel
embeds in the next power of two.  Both compute the same Toeplitz product (the
embedding length only has to be >= 2n - 1), so results agree to roundoff.
"""
import numpy as np
import scipy.linalg as la

from .bttb import BTTB

_EPS = np.finfo(np.float64).eps


class Toeplitz(BTTB):
    def __init__(self, top):
        top = np.asarray(top)
        if top.ndim != 1:
            raise ValueError('top shape {} is not 1D'.format(top.shape))
        if top.size == 0:
            raise ValueError('top is empty')
        super().__init__(top, (top.size,))

    def as_numpy(self):
        return la.toeplitz(self.top)

    def upper_eig_bound(self):
        """Gershgorin: the largest absolute row sum, in O(n) (reference
        toeplitz.py:69-85)."""
        a = np.abs(self.top)
        rows = a.copy()
        rows[0] = a.sum()
        rows[1:] -= a[:0:-1]
        return np.add.accumulate(rows).max() * (1 + _EPS * a.size)

    def __str__(self):
        if self.top.size > 10:
            return 'Toeplitz size {}'.format(self.top.size)
        return 'Toeplitz {}'.format(self.top)
