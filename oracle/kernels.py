"""Oracle: stationary kernels and a paramz-free kernel specification.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The reference's FunctionalKernel / RBF / Matern32 / StdPeriodic need paramz
(absent here and on the GPU box); these plain classes expose the same
duck-typed surface the hot path consumes (reference
runlmc/lmc/functional_kernel.py:225-302) so that the reference's own
gen_grid_kernel / ApproxLMCLikelihood can be driven by them when golden
vectors are generated, and so the oracle can be driven identically.
"""
import numpy as np


class RBFSpec:
    """k(r) = exp(-r^2 gamma / 2) (reference runlmc/kern/rbf.py:39-40,51-54)."""
    n_params = 1

    def __init__(self, inv_lengthscale=1.0, active_dims=None):
        self.inv_lengthscale = float(inv_lengthscale)
        self.active_dims = active_dims

    def from_dist(self, d):
        return np.exp(-0.5 * np.square(d) * self.inv_lengthscale)

    def kernel_gradient(self, d):
        sq = np.square(d)
        return [np.exp(-0.5 * sq * self.inv_lengthscale) * -0.5 * sq]


class Matern32Spec:
    """(1 + s) exp(-s), s = sqrt(3) gamma r (reference
    runlmc/kern/matern32.py:39-41,52-57)."""
    n_params = 1

    def __init__(self, inv_lengthscale=1.0, active_dims=None):
        self.inv_lengthscale = float(inv_lengthscale)
        self.active_dims = active_dims

    def from_dist(self, d):
        s = d * np.sqrt(3) * self.inv_lengthscale
        return (1 + s) * np.exp(-s)

    def kernel_gradient(self, d):
        s = d * np.sqrt(3) * self.inv_lengthscale
        ds = d * np.sqrt(3)
        e = np.exp(-s)
        return [(1 + s) * (e * -ds) + ds * e]


class StdPeriodicSpec:
    """exp(-gamma sin^2(pi r / T) / 2) (reference
    runlmc/kern/std_periodic.py:44-48,60-67)."""
    n_params = 2

    def __init__(self, inv_lengthscale=1.0, period=1.0, active_dims=None):
        self.inv_lengthscale = float(inv_lengthscale)
        self.period = float(period)
        self.active_dims = active_dims

    def from_dist(self, d):
        if np.log(self.period) < -200:
            return np.nan
        s = np.sin((np.pi / self.period) * d)
        return np.exp(-0.5 * np.square(s) * self.inv_lengthscale)

    def kernel_gradient(self, d):
        scaled = np.pi / self.period * d
        s = np.sin(scaled)
        ds = np.cos(scaled) * scaled
        ds = ds * (-1 / self.period * self.inv_lengthscale)
        sq = np.square(s)
        e = np.exp(-0.5 * sq * self.inv_lengthscale)
        return [e * -0.5 * sq, e * -1 * s * ds]


class ScaledSpec:
    """scale * k(r) with `scale` a differentiable parameter appended last
    (reference runlmc/kern/scaled.py:13-37)."""

    def __init__(self, k, scale=1.0):
        self.k = k
        self.scale = float(scale)
        self.active_dims = k.active_dims
        self.n_params = k.n_params + 1

    def from_dist(self, d):
        return self.scale * self.k.from_dist(d)

    def kernel_gradient(self, d):
        return [self.scale * g for g in self.k.kernel_gradient(d)] + [self.k.from_dist(d)]


class KernelSpec:
    """Paramz-free stand-in for the reference FunctionalKernel, LMC kernels
    only or LMC + SLFM + independent (reference functional_kernel.py:86-302).

    :param D: number of outputs
    :param kernels: list of kernel objects, ordered lmc, slfm, indep
    :param coreg_vecs: list of (R_q x D) arrays
    :param coreg_diags: list of (D,) arrays
    :param noise: (D,) array
    :param num_lmc, num_slfm: how many of `kernels` are LMC / SLFM kernels
    """

    def __init__(self, D, kernels, coreg_vecs, coreg_diags, noise,
                 num_lmc=None, num_slfm=0):
        self.D = int(D)
        self._kernels = list(kernels)
        self.coreg_vecs = [np.atleast_2d(np.asarray(a, dtype=np.float64))
                           for a in coreg_vecs]
        self.coreg_diags = [np.asarray(k, dtype=np.float64)
                            for k in coreg_diags]
        self.noise = np.asarray(noise, dtype=np.float64)
        self._num_lmc = len(kernels) if num_lmc is None else int(num_lmc)
        self._num_slfm = int(num_slfm)
        self.P = None
        self.active_dims = {}
        self.num_lmc, self.num_slfm, self.num_indep = {}, {}, {}

    @property
    def Q(self):
        return len(self._kernels)

    def set_input_dim(self, P):
        # functional_kernel.py:144-167
        self.P = P
        everything = tuple(range(P))
        for i, k in enumerate(self._kernels):
            k.active_dims = (everything if k.active_dims is None
                             else tuple(sorted(k.active_dims)))
            self.active_dims.setdefault(k.active_dims, []).append(i)
            if i < self._num_lmc:
                bucket = self.num_lmc
            elif i < self._num_lmc + self._num_slfm:
                bucket = self.num_slfm
            else:
                bucket = self.num_indep
            bucket[k.active_dims] = bucket.get(k.active_dims, 0) + 1
        for bucket in (self.num_lmc, self.num_slfm, self.num_indep):
            for ad in self.active_dims:
                bucket.setdefault(ad, 0)

    def total_rank(self, active_dim):
        # functional_kernel.py:225-232
        return sum(len(self.coreg_vecs[q]) for q in self.active_dims[active_dim]
                   if q < self._num_lmc + self._num_slfm)

    def eval_kernels(self, dists):
        return [k.from_dist(dists[k.active_dims]) for k in self._kernels]

    def eval_kernels_fixed_dim(self, dists, active_dim):
        return np.array([self._kernels[q].from_dist(dists)
                         for q in self.active_dims[active_dim]])

    def eval_kernel_gradients(self, dists):
        return [k.kernel_gradient(dists[k.active_dims])
                for k in self._kernels]

    def coreg_mats(self, active_dim=None):
        cv, cd = self.coreg_vecs, self.coreg_diags
        if active_dim is not None:
            idx = self.active_dims[active_dim]
            cv, cd = [cv[i] for i in idx], [cd[i] for i in idx]
        return [a.T.dot(a) + np.diag(k) for a, k in zip(cv, cd)]

    def get_active_dims(self, q):
        return self._kernels[q].active_dims

    def filter_non_indep_idxs(self, idxs):
        lim = self._num_lmc + self._num_slfm
        return [i for i in idxs if i < lim]
