"""Oracle: cubic-convolution interpolation weights and grid generation.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
Restates runlmc/approx/interpolation.py (1-D path) with plain loops.
"""
import numpy as np
import scipy.sparse


def keys_cubic(s):
    """Keys' cubic convolution kernel on |s| <= 2 (reference
    interpolation.py:21-53)."""
    s = np.fabs(np.asarray(s, dtype=np.float64))
    if np.any(s > 2):
        raise ValueError('only absolute values <= 2 allowed')
    near = ((1.5 * s - 2.5) * s) * s + 1
    far = ((-0.5 * s + 2.5) * s - 4) * s + 2
    return np.where(s <= 1, near, far)


def cubic_rows(grid, samples):
    """Dense (n x m) cubic interpolation matrix: four taps per sample at
    floor-index + {2, 1, 0, -1}... (reference interpolation.py:56-116: the
    loop runs conv_idx = -2..1, column = idx_of_closest - conv_idx clamped to
    [0, m-1], weight = u(frac + conv_idx); clamped duplicates are SUMMED)."""
    grid = np.asarray(grid, dtype=np.float64)
    samples = np.asarray(samples, dtype=np.float64).ravel()
    m = len(grid)
    if m < 4:
        raise ValueError('grid size must be >= 4')
    out = np.zeros((samples.size, m))
    step = grid[1] - grid[0]
    for i, s in enumerate(samples):
        f = (s - grid[0]) / step
        base = np.floor(f)
        frac = f - base
        for shift in (-2, -1, 0, 1):
            col = int(min(max(base - shift, 0), m - 1))
            out[i, col] += keys_cubic(frac + shift)
    return out


def multi_interp(Xs, grid):
    """Block-diagonal (sum n_d) x (D m) CSR interpolant (reference
    interpolation.py:119-176)."""
    blocks = [scipy.sparse.csr_matrix(cubic_rows(grid, np.asarray(X).ravel()))
              for X in Xs]
    return scipy.sparse.block_diag(blocks, format='csr')


def auto_grid_1d(Xs, m=None):
    """Grid that covers every input with two spare cells each side and
    m + 4 points (reference interpolation.py:179-215, 1-D, lo=hi=None)."""
    lo = min(float(np.min(X)) for X in Xs)
    hi = max(float(np.max(X)) for X in Xs)
    if m is None:
        m = sum(len(X) for X in Xs) // len(Xs)
    delta = (hi - lo) / m
    return np.linspace(lo - 2 * delta, hi + 2 * delta, int(m) + 4)


def bicubic_rows(gridx, gridy, samples):
    """Dense n x (mx*my) bicubic interpolation matrix: the tensor product of
    the two 1-D cubic rows of each sample (reference interpolation.py:218-328:
    interpolate along x at four grid rows, then along y); index ix*my + iy."""
    samples = np.asarray(samples, dtype=np.float64)
    Rx = cubic_rows(gridx, samples[:, 0])
    Ry = cubic_rows(gridy, samples[:, 1])
    return np.einsum('ni,nj->nij', Rx, Ry).reshape(len(samples), -1)
