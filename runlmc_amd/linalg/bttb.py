"""Symmetric (block-)Toeplitz operator on the device.

Mirror of reference runlmc/linalg/bttb.py:22-155.  The circulant embedding,
its spectrum and pad->FFT->multiply->IFFT->crop all happen in the HIP library
(rl_gridop_* with D = 1, B = [[1]]).  The embedding length is the reference's
``next pow2 >= 2m`` (bttb.py:16-19), floored at 16.

One- and two-dimensional grids have a device path (a 2-D BTTB is embedded in a
two-dimensional circulant, one power-of-two length per axis, exactly as the
reference's rfftn over ``sizes``); three or more dimensions raise
NotImplementedError.
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
        if len(self._sizes) > 2:
            raise NotImplementedError(
                'device BTTB supports 1-D and 2-D grids; got sizes {}'
                .format(self._sizes))
        self._dev = None

    def _device_op(self):
        if self._dev is None:
            op = GridOp(1, self.top.size, 1, sizes=self._sizes)
            op.set_dense(self.top.reshape(1, -1), np.ones((1, 1, 1)))
            self._dev = op
        return self._dev

    def matvec(self, x):
        x = check_vector(x, self.shape[1])
        return self._device_op().matmat_host(x.astype(np.float64))

    def matmat(self, X):
        X = check_block(X, self.shape[1])
        rows = np.ascontiguousarray(X.T, dtype=np.float64)
        if rows.shape[0] == 0:
            return np.empty((self.shape[0], 0))
        return self._device_op().matmat_host(rows).T

    def as_numpy(self):
        return _dense_bttb(self.top, self._sizes)

    def __getstate__(self):
        state = super().__getstate__()
        state['_dev'] = None     # device handles do not travel
        return state

    def __str__(self):
        if self.top.size > 50:
            return 'BTTB on grid shape {}'.format(self._sizes)
        return 'BTTB on grid\n{}'.format(self.top.reshape(self._sizes))
