"""Synthetic LMC problems with the reference benchmark's recipe.

Restates the input generation of reference benchmarks/benchlib/bench.py:105-164
(seed 1234, truncated-normal A_q, inverse-gamma kappa_q and noise, RBF kernels
with inverse lengthscales logspace(0, 1, Q), inputs/outputs U(0, 1), n_o = m
points per output, autogrid => m + 4 grid points, cubic interpolant) so that
the CPU baseline and the GPU path see identical data.
"""
import numpy as np
import scipy.stats

from ..approx.interpolation import autogrid, multi_interpolant
from ..kern.stationary import RBF
from ..lmc.functional_kernel import FunctionalKernel

CONFIGS = {
    # name: (D, Q, R, m, n_probes)      
    'c1': (2, 2, 1, 82, 15),
    'c2': (4, 3, 1, 5000, 16),
    'c5': (10, 5, 1, 100000, 128),
}


class SynthProblem:
    """Plain container: parameters, data, grid and interpolants."""


def make_problem(D, Q, R, m, eps=0.1, seed=1234):
    """n_o = m inputs per output; returns a SynthProblem."""
    np.random.seed(seed)
    p = SynthProblem()
    p.D, p.Q, p.R, p.n_o = D, Q, R, m
    p.coreg_vecs = scipy.stats.truncnorm(-1, 1).rvs(size=(Q, R, D))
    p.coreg_diags = np.reciprocal(np.random.gamma(shape=1, scale=1, size=(Q, D)))
    p.noise = np.reciprocal(np.random.gamma(shape=(1 + (1 / eps)), scale=1, size=D))
    p.inv_lengthscales = np.logspace(0, 1, Q)
    Xs, Ys = np.random.rand(2, D, m)
    p.Xs = [x.reshape(-1, 1) for x in Xs]
    p.Ys = [y for y in Ys]
    p.lens = [m] * D
    p.n = D * m
    p.grid = autogrid(p.Xs, lo=None, hi=None, m=None)[0]
    p.grid_dists = p.grid - p.grid[0]
    p.m = len(p.grid)
    p.W = multi_interpolant(p.Xs, p.grid)
    p.WT = p.W.transpose().tocsr()
    p.WT.sort_indices()
    p.WT.indices = p.WT.indices.astype(np.int32)
    p.WT.indptr = p.WT.indptr.astype(np.int32)
    p.y = np.hstack(p.Ys)
    return p


def functional_kernel(p):
    """The package's FunctionalKernel for a SynthProblem (all LMC kernels)."""
    fk = FunctionalKernel(D=p.D,
                          lmc_kernels=[RBF(g) for g in p.inv_lengthscales],
                          lmc_ranks=[p.R] * p.Q)
    fk.coreg_vecs = list(p.coreg_vecs)
    fk.coreg_diags = list(p.coreg_diags)
    fk.noise = p.noise
    fk.set_input_dim(1)
    return fk


def tops(p):
    """k_q(grid distances), shape (Q, m)."""
    d = p.grid_dists
    return np.array([np.exp(-0.5 * np.square(d) * g) for g in p.inv_lengthscales])


def algorithmic_bytes_grid_mvm(D, Q, m, L, nvec):
    """read x, write y, read the Q real spectra once."""
    return 8 * (2 * D * m * nvec + Q * (L // 2 + 1))
