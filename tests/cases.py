"""Shared helpers for the parity tests: rebuild a stored LMC case.

The .npz fixtures under tests/golden/ hold INPUTS (seeded, stored, never
re-drawn) and the reference's OUTPUTS for them (tests/golden/make_golden.py).
"""
import os

import numpy as np
import scipy.sparse

from oracle.kernels import (KernelSpec, RBFSpec, Matern32Spec, StdPeriodicSpec,
                            ScaledSpec)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def kernel_from_desc(desc):
    parts = str(desc).split(';')
    kind, vals = parts[0], [float(v) for v in parts[1:]]
    if kind == 'rbf':
        return RBFSpec(*vals)
    if kind == 'matern':
        return Matern32Spec(*vals)
    if kind == 'periodic':
        return StdPeriodicSpec(*vals)
    if kind == 'scaled_rbf':
        return ScaledSpec(RBFSpec(vals[0]), vals[1])
    raise ValueError(kind)


class Case:
    """One golden LMC case: inputs as attributes, reference outputs in .g"""

    def __init__(self, name):
        g = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.g = g
        self.name = name
        self.D, self.Q, self.m = int(g['D']), int(g['Q']), int(g['m'])
        self.num_lmc = int(g['num_lmc']) if 'num_lmc' in g else self.Q
        self.num_slfm = int(g['num_slfm']) if 'num_slfm' in g else 0
        self.P = int(g['P']) if 'P' in g else 1          # input dimension
        self.ad = tuple(range(self.P))                   # active-dimension key
        self.lens = [int(v) for v in g['lens']]
        self.n = sum(self.lens)
        self.grid_dists = g['grid_dists']
        self.noise = g['noise']
        self.y = g['y']
        self.coreg_vecs = [g[f'A{q}'] for q in range(self.Q)]
        self.coreg_diags = [g[f'kappa{q}'] for q in range(self.Q)]
        self.kdesc = [str(k) for k in g['kdesc']]
        self.W = scipy.sparse.csr_matrix(
            (g['W_data'], g['W_indices'], g['W_indptr']),
            shape=(self.n, self.D * self.m))
        self.WT = scipy.sparse.csr_matrix(
            (g['WT_data'], g['WT_indices'], g['WT_indptr']),
            shape=(self.D * self.m, self.n))
        self.Xs = [g[f'X{d}'] for d in range(self.D)]
        self.grid_axes = ([g['axis_x'], g['axis_y']] if self.P == 2 else
                          ([g['grid']] if 'grid' in g else None))
        self.Ys = np.split(self.y, np.cumsum(self.lens)[:-1])
        self.rs = g['rs']
        self.tops = g['tops']

    def spec(self):
        sp = KernelSpec(self.D, [kernel_from_desc(k) for k in self.kdesc],
                        self.coreg_vecs, self.coreg_diags, self.noise,
                        num_lmc=self.num_lmc, num_slfm=self.num_slfm)
        sp.set_input_dim(self.P)
        return sp

    def dtops(self):
        cnt = self.g['dtops_count']
        return [[self.g[f'dtop{q}_{p}'] for p in range(int(cnt[q]))]
                for q in range(self.Q)]


ALL_CASES = ['lmc_small', 'lmc_c1', 'lmc_q1', 'lmc_mid', 'lmc_2d']
DENSE_CASES = ['lmc_small', 'lmc_c1', 'lmc_q1', 'lmc_2d']
# every top row smooth over the grid (round 6): the operators the device inverts directly;
# dense alpha / K~^-1 r / log det / gradients from the reference's own dense Cholesky + loops
SMOOTH_CASES = ['lmc_smooth']
# the reference's real-data workloads (BASELINE configs 3 and 4)
DATASET_CASES = ['fx2007', 'weather']
