"""Structured kernel interpolation W K W^T (mirror of reference
runlmc/approx/ski.py:8-23).

Generic form: composes any grid operator K with the CSR interpolants on the
host.  The LMC hot path never goes through here -- GridKernel
(runlmc_amd/lmc/grid_kernel.py) keeps W, W^T and K_UU on the device and runs
the whole product there."""
from ..linalg.composition import Composition
from ..linalg.matrix import Matrix


class SKI(Composition):
    def __init__(self, K, W, WT):
        self.W = W
        self.K = K
        self.WT = WT
        super().__init__([Matrix.wrap(W.shape, W.dot), K,
                          Matrix.wrap(WT.shape, WT.dot)])

    def as_numpy(self):
        half = self.W.dot(self.K.as_numpy().T)      # W K^T
        return self.W.dot(half.T)                   # W (W K^T)^T = W K W^T

    def upper_eig_bound(self):
        # the reference's version refers to an attribute that is never set
        # (ski.py:22-23); use the obvious bound instead
        return self.K.upper_eig_bound() * self.shape[0] / self.K.shape[0]
