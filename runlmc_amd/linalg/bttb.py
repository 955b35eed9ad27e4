"""Symmetric (block-)Toeplitz operator on the device.

Mirror of reference runlmc/linalg/bttb.py:22-155.  The circulant embedding,
its spectrum and pad->FFT->multiply->IFFT->crop all happen in the HIP library
(rl_gridop_* with D = 1, B = [[1]]).  The embedding length is the reference's
``next pow2 >= 2m`` (bttb.py:16-19), floored at 16.

One- and two-dimensional grids have a device path of their own (a 2-D BTTB is
embedded in a two-dimensional circulant, one power-of-two length per axis,
exactly as the reference's rfftn over ``sizes``).  Three or more dimensions
(reference bttb.py:110-148 takes any number; its SKI path never builds them --
interpolation.py has cubic and bicubic weights only) are reduced to those: an
N-D BTTB is block-Toeplitz over its leading axes with 2-D BTTB blocks,
``y_i = sum_j T_{|i - j|} x_j``, so the product is one batched 2-D device
product per leading lag (P = prod(sizes[:-2]) of them, each over the P slabs of
the input) and a sum -- O(P^2) two-dimensional products instead of one N-D
transform, which is the right trade for the small leading extents such grids
have.
"""
import numpy as np

from .matrix import Matrix, check_vector, check_block
from .._native import GridOp


def _dense_bttb(top, sizes):
    """Dense matrix whose (i, j) entry is top at the per-axis |i_p - j_p|."""
    grid = np.indices(sizes).reshape(len(sizes), -1)
    lag = np.abs(grid[:, :, None] - grid[:, None, :])
    return top.reshape(sizes)[tuple(lag)]


class BTTB(Matrix):
    def __init__(self, top, sizes):
        top = np.asarray(top)
        sizes = np.asarray(sizes)
        if top.ndim != 1:
            raise ValueError('top shape {} is not 1D'.format(top.shape))
        if top.size == 0:
            raise ValueError('top is empty')
        if sizes.ndim != 1:
            raise ValueError('sizes shape {} is not 1D'.format(sizes.shape))
        if int(np.prod(sizes)) != top.size:
            raise ValueError("sizes {} don't match grid size {}"
                             .format(sizes, top.size))
        super().__init__(top.size, top.size)
        # unsafe casts (e.g. complex) raise TypeError, as in the reference
        self.top = top.astype('float64', casting='safe')
        self._sizes = tuple(int(s) for s in sizes)
        self._dev = None
        self._lead = self._sizes[:-2] if len(self._sizes) > 2 else ()
        self._pairs = None

    def _device_op(self):
        if self._dev is None:
            if self._lead:
                # one top row per leading lag, each a 2-D BTTB over the last two axes
                P = int(np.prod(self._lead))
                inner = self._sizes[-2:]
                op = GridOp(1, inner[0] * inner[1], P, sizes=inner)
                op.set_dense(self.top.reshape(P, -1), np.ones((P, 1, 1)))
                grid = np.indices(self._lead).reshape(len(self._lead), P)
                lag = np.abs(grid[:, :, None] - grid[:, None, :])
                lagmat = np.ravel_multi_index(tuple(lag), self._lead)     # (P, P): lag of (i, j)
                self._pairs = [np.nonzero(lagmat == t) for t in range(P)]
            else:
                op = GridOp(1, self.top.size, 1, sizes=self._sizes)
                op.set_dense(self.top.reshape(1, -1), np.ones((1, 1, 1)))
            self._dev = op
        return self._dev

    def _rows(self, rows):
        """rows: (k, n) float64, one vector per row -> (k, n)."""
        op = self._device_op()
        if not self._lead:
            return op.matmat_host(rows)
        k = rows.shape[0]
        P = int(np.prod(self._lead))
        m2 = self.top.size // P
        slabs = np.ascontiguousarray(rows.reshape(k * P, m2))
        y = np.zeros((P, k, m2))
        for t in range(P):
            # T_t applied to every slab of every vector, then added into the
            # slabs i with |i - j| = t
            Z = op.matmat_host(slabs, top=t).reshape(k, P, m2).transpose(1, 0, 2)
            i_idx, j_idx = self._pairs[t]
            np.add.at(y, i_idx, Z[j_idx])
        return np.ascontiguousarray(y.transpose(1, 0, 2)).reshape(k, -1)

    def matvec(self, x):
        x = check_vector(x, self.shape[1])
        return self._rows(np.ascontiguousarray(x, dtype=np.float64)[None, :])[0]

    def matmat(self, X):
        X = check_block(X, self.shape[1])
        rows = np.ascontiguousarray(X.T, dtype=np.float64)
        if rows.shape[0] == 0:
            return np.empty((self.shape[0], 0))
        return self._rows(rows).T

    def as_numpy(self):
        return _dense_bttb(self.top, self._sizes)

    def __getstate__(self):
        state = super().__getstate__()
        state['_dev'] = None     # device handles do not travel
        state['_pairs'] = None
        return state

    def __str__(self):
        if self.top.size > 50:
            return 'BTTB on grid shape {}'.format(self._sizes)
        return 'BTTB on grid\n{}'.format(self.top.reshape(self._sizes))
