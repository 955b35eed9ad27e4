"""Synthetic LMC problems with the reference benchmark's recipe.

Restates the input generation of reference benchmarks/benchlib/bench.py:105-164
(seed 1234, truncated-normal A_q, inverse-gamma kappa_q and noise, inputs/outputs
U(0, 1), n_o = m points per output, autogrid => m + 4 grid points, cubic
interpolant) and its four kernel families (bench.py:94,284-297: 'rbf',
'periodic', 'matern', 'mix') so that the CPU baseline and the GPU path see
identical data.
"""
import numpy as np
import scipy.stats

from ..approx.interpolation import autogrid, multi_interpolant
from ..kern.stationary import RBF, Matern32, StdPeriodic
from ..lmc.functional_kernel import FunctionalKernel

CONFIGS = {
    # name: (D, Q, R, m, n_probes)      
    'c1': (2, 2, 1, 82, 15),
    'c2': (4, 3, 1, 5000, 16),
    'c5': (10, 5, 1, 100000, 128),
}


KERN_FAMILIES = ('rbf', 'periodic', 'matern', 'mix')


def kernel_family(Q, kern='rbf'):
    """The reference benchmark's kernels (benchmarks/benchlib/bench.py:284-297,
    gen_kernels) as plain descriptions ('rbf', gamma) / ('periodic', gamma,
    period) / ('matern', gamma):
      rbf       inverse length scales logspace(0, 1, Q)
      periodic  inverse length scale 1, periods logspace(0, 1, Q)
      matern    Matern-3/2, inverse length scales logspace(0, 1, Q)
      mix       rbf, periodic, matern at logspace(0, 1, max(Q // 3, 1)), cut to Q
                or padded with rbf(1)."""
    gam = np.logspace(0, 1, Q)
    if kern == 'rbf':
        return [('rbf', float(g)) for g in gam]
    if kern == 'periodic':
        return [('periodic', 1.0, float(g)) for g in gam]
    if kern == 'matern':
        return [('matern', float(g)) for g in gam]
    if kern == 'mix':
        mix = []
        for g in np.logspace(0, 1, max(Q // 3, 1)):
            mix += [('rbf', float(g)), ('periodic', 1.0, float(g)), ('matern', float(g))]
        mix = mix[:Q]
        mix += [('rbf', 1.0)] * (Q - len(mix))
        return mix
    raise ValueError('kern must be one of %s' % (KERN_FAMILIES,))


def kernel_objects(desc, rbf=RBF, periodic=StdPeriodic, matern=Matern32):
    """Kernel objects for a kernel_family() description (the oracle passes its
    own paramz-free classes)."""
    make = {'rbf': rbf, 'periodic': periodic, 'matern': matern}
    return [make[d[0]](*d[1:]) for d in desc]


class SynthProblem:
    """Plain container: parameters, data, grid and interpolants."""


def make_problem(D, Q, R, m, eps=0.1, seed=1234, kern='rbf'):
    """n_o = m inputs per output; returns a SynthProblem.  (The kernels draw
    nothing from the generator: every family sees the same data.)"""
    np.random.seed(seed)
    p = SynthProblem()
    p.D, p.Q, p.R, p.n_o = D, Q, R, m
    p.kern = kern
    p.kern_desc = kernel_family(Q, kern)
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
                          lmc_kernels=kernel_objects(p.kern_desc),
                          lmc_ranks=[p.R] * p.Q)
    fk.coreg_vecs = list(p.coreg_vecs)
    fk.coreg_diags = list(p.coreg_diags)
    fk.noise = p.noise
    fk.set_input_dim(1)
    return fk


def tops(p):
    """k_q(grid distances), shape (Q, m)."""
    return np.array([k.from_dist(p.grid_dists) for k in kernel_objects(p.kern_desc)])


def algorithmic_bytes_grid_mvm(D, Q, m, L, nvec):
    """read x, write y, read the Q real spectra once."""
    return 8 * (2 * D * m * nvec + Q * (L // 2 + 1))
