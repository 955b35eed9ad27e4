"""Sparse cubic-convolution interpolation weights and grid generation.

One-time host-side setup (NumPy/SciPy), mirroring reference
runlmc/approx/interpolation.py: the CSR matrices built here are uploaded once
to the device (rl_ski_create) and only their products run on the GPU.
"""
import logging

import numpy as np
import scipy.sparse

_LOG = logging.getLogger(__name__)


def cubic_kernel(x):
    """Keys' cubic convolution kernel on |x| <= 2 (reference
    interpolation.py:21-53)."""
    x = np.fabs(np.asarray(x, dtype=np.float64))
    if np.any(x > 2):
        raise ValueError('only absolute values <= 2 allowed')
    inner = ((1.5 * x - 2.5) * x) * x + 1
    outer = ((-0.5 * x + 2.5) * x - 4) * x + 2
    return np.where(x <= 1, inner, outer)


def interp_cubic(grid, samples):
    """n x m CSR matrix M with four taps per row such that M f(grid) ~
    f(samples) (reference interpolation.py:56-116).  Taps that would fall
    off the grid are clamped to the end points and their weights summed."""
    grid = np.asarray(grid)
    samples = np.asarray(samples)
    m = len(grid)
    n = samples.size
    if n == 0:
        return scipy.sparse.csr_matrix((0, m), dtype=float)
    if grid.ndim != 1:
        raise ValueError('grid dim {} should be 1'.format(grid.ndim))
    if samples.ndim != 1:
        raise ValueError('samples dim {} should be 1'.format(samples.ndim))
    if m < 4:
        raise ValueError('grid size {} must be >=4'.format(m))
    if samples.min() <= grid[0] or samples.max() >= grid[-1]:
        _LOG.warning('range of samples [%f, %f] outside grid range [%f, %f]',
                     samples.min(), samples.max(), grid[0], grid[-1])
    step = grid[1] - grid[0]
    pos = (samples - grid[0]) / step
    left = np.floor(pos)
    frac = pos - left
    shifts = np.array([-2, -1, 0, 1])
    cols = np.clip(left[:, None] - shifts[None, :], 0, m - 1).astype(np.int64)
    vals = cubic_kernel(frac[:, None] + shifts[None, :])
    rows = np.repeat(np.arange(n), 4)
    M = scipy.sparse.coo_matrix((vals.ravel(), (rows, cols.ravel())),
                                shape=(n, m))
    return M.tocsr()       # duplicates (clamped taps) are summed


def interp_bicubic(gridx, gridy, samples):
    """n x (mx*my) CSR matrix with 16 taps per row: cubic interpolation along
    x at the four grid rows around each sample, then along y (reference
    interpolation.py:218-328).  Grid index = ix * my + iy."""
    gridx, gridy = np.asarray(gridx), np.asarray(gridy)
    samples = np.asarray(samples)
    mx, my = gridx.size, gridy.size
    n = samples.shape[0]
    if n == 0:
        return scipy.sparse.csr_matrix((0, mx * my), dtype=float)
    for name, grid in (('gridx', gridx), ('gridy', gridy)):
        if grid.ndim != 1:
            raise ValueError('{} dim {} should be 1'.format(name, grid.ndim))
        if grid.size < 4:
            raise ValueError('grid size {} must be >=4'.format(grid.size))
    if samples.ndim != 2 or samples.shape[1] != 2:
        raise ValueError('expecting 2d samples, got shape {}'.format(samples.shape))
    for axis, grid in ((0, gridx), (1, gridy)):
        col = samples[:, axis]
        if col.min() <= grid[0] or col.max() >= grid[-1]:
            _LOG.warning('%s range of samples [%f, %f] outside grid range [%f, %f]',
                         'xy'[axis], col.min(), col.max(), grid[0], grid[-1])
    shifts = np.array([-2, -1, 0, 1])

    def taps(grid, coords):
        pos = (coords - grid[0]) / (grid[1] - grid[0])
        left = np.floor(pos)
        idx = np.clip(left[:, None] - shifts[None, :], 0, len(grid) - 1).astype(np.int64)
        return idx, cubic_kernel((pos - left)[:, None] + shifts[None, :])

    ix, wx = taps(gridx, samples[:, 0])          # (n, 4) each
    iy, wy = taps(gridy, samples[:, 1])
    cols = (ix[:, :, None] * my + iy[:, None, :]).reshape(n, 16)
    vals = (wx[:, :, None] * wy[:, None, :]).reshape(n, 16)
    rows = np.repeat(np.arange(n), 16)
    return scipy.sparse.coo_matrix((vals.ravel(), (rows, cols.ravel())),
                                   shape=(n, mx * my)).tocsr()


def multi_interpolant(Xs, *inducing_grids):
    """Block-diagonal interpolant over all outputs: (sum_d n_d) x (D m) CSR
    with int32 indices (reference interpolation.py:119-176)."""
    if Xs[0].ndim == 1 or Xs[0].shape[1] == 1:
        blocks = [interp_cubic(inducing_grids[0], np.asarray(X).ravel())
                  for X in Xs]
    elif Xs[0].shape[1] == 2:
        blocks = [interp_bicubic(inducing_grids[0], inducing_grids[1], np.asarray(X))
                  for X in Xs]
    else:
        raise NotImplementedError('inputs of more than two dimensions')
    W = scipy.sparse.block_diag(blocks, format='csr')
    W.sort_indices()
    W.indices = W.indices.astype(np.int32)
    W.indptr = W.indptr.astype(np.int32)
    return W


def autogrid(Xs, lo, hi, m):
    """Equally spaced grid per input dimension that leaves two spare cells on
    both sides of the data and therefore has m + 4 points (reference
    interpolation.py:179-215).  lo / hi / m may be None."""
    P = Xs[0].shape[1]
    for name, val in (('lo', lo), ('hi', hi), ('m', m)):
        assert val is None or len(val) == P, (name, P, val)
    data_lo = np.min([X.min(axis=0) for X in Xs], axis=0)
    data_hi = np.max([X.max(axis=0) for X in Xs], axis=0)
    if m is None:
        m = np.ones(P) * (sum(len(X) for X in Xs) // len(Xs))
    else:
        m = np.array(m, dtype=float)
    lo = (data_lo if lo is None else np.minimum(lo, data_lo)).astype(float)
    hi = (data_hi if hi is None else np.maximum(hi, data_hi)).astype(float)
    delta = (hi - lo) / m
    return [np.linspace(l - 2 * d, h + 2 * d, int(mm) + 4)
            for l, h, d, mm in zip(lo, hi, delta, m)]
