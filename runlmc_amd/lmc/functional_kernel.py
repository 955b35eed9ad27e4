"""LMC kernel specification (paramz-free mirror of reference
runlmc/lmc/functional_kernel.py:17-302).

Same constructor and the same duck-typed surface the hot path consumes:
``D, Q, active_dims, num_lmc/num_slfm/num_indep, total_rank, eval_kernels,
eval_kernels_fixed_dim, eval_kernel_gradients, coreg_vecs, coreg_diags,
coreg_mats, noise, get_active_dims, filter_non_indep_idxs`` and the gradient
sink ``update_gradient(grads)``.  Parameters are plain arrays; gradients land
in ``coreg_vec_grads``, ``coreg_diag_grads``, ``noise_grad`` and on each
kernel's ``gradient``."""
import numpy as np
import scipy.stats


class FunctionalKernel:
    def __init__(self, D=None, lmc_kernels=None, lmc_ranks=None,
                 slfm_kernels=None, indep_gp=None, indep_gp_index=None,
                 name='kern'):
        self.name = name
        if not D:
            raise ValueError('D should be specified')
        self.D = int(D)
        lmc_kernels = list(lmc_kernels or [])
        lmc_ranks = list(lmc_ranks or [])
        slfm_kernels = list(slfm_kernels or [])
        indep_gp = list(indep_gp or [])
        if not lmc_kernels and not slfm_kernels and not indep_gp:
            raise ValueError('Number of kernels should be >0')
        if len(lmc_kernels) != len(lmc_ranks):
            raise ValueError('# LMC kernels should equal # LMC ranks')
        if any(r <= 0 for r in lmc_ranks):
            raise ValueError('LMC ranks not positive')
        indep_gp_index = list(indep_gp_index if indep_gp_index is not None
                              else range(len(indep_gp)))
        if len(indep_gp) != len(indep_gp_index):
            raise ValueError('indep GP number of kernels should match indices')

        self._kernels = lmc_kernels + slfm_kernels + indep_gp
        self._num_lmc = len(lmc_kernels)
        self._num_slfm = len(slfm_kernels)

        draw = scipy.stats.truncnorm(-1, 1).rvs
        self._coreg_vecs = (
            [draw(size=(r, self.D)) for r in lmc_ranks] +
            [draw(size=(1, self.D)) for _ in slfm_kernels] +
            [np.zeros((1, self.D)) for _ in indep_gp])
        # kappa: ones for LMC terms, fixed zeros for SLFM terms, fixed
        # indicator of the owning output for independent GPs
        self._coreg_diags = (
            [np.ones(self.D) for _ in lmc_kernels] +
            [np.zeros(self.D) for _ in slfm_kernels])
        for d in indep_gp_index:
            e = np.zeros(self.D)
            e[d] = 1
            self._coreg_diags.append(e)
        self._noise = 0.1 * np.ones(self.D)

        self.coreg_vec_grads = [np.zeros_like(a) for a in self._coreg_vecs]
        self.coreg_diag_grads = [np.zeros_like(k) for k in self._coreg_diags]
        self.noise_grad = np.zeros(self.D)

        self.P = None
        self.active_dims = {}
        self.num_lmc, self.num_slfm, self.num_indep = {}, {}, {}

    # -- structure ---------------------------------------------------------
    def set_input_dim(self, P):
        if self.P == P:
            return
        if self.P is not None:
            raise ValueError('Cannot set input dimension twice')
        self.P = P
        everything = tuple(range(P))
        for q, k in enumerate(self._kernels):
            k.active_dims = (everything if k.active_dims is None
                             else tuple(sorted(k.active_dims)))
            self.active_dims.setdefault(k.active_dims, []).append(q)
            if q < self._num_lmc:
                counter = self.num_lmc
            elif q < self._num_lmc + self._num_slfm:
                counter = self.num_slfm
            else:
                counter = self.num_indep
            counter[k.active_dims] = counter.get(k.active_dims, 0) + 1
        for counter in (self.num_lmc, self.num_slfm, self.num_indep):
            for ad in self.active_dims:
                counter.setdefault(ad, 0)

    @property
    def Q(self):
        return len(self._kernels)

    @property
    def kernels(self):
        return list(self._kernels)

    def total_rank(self, active_dim):
        assert self.P
        lim = self._num_lmc + self._num_slfm
        return sum(len(self._coreg_vecs[q])
                   for q in self.active_dims[active_dim] if q < lim)

    def get_active_dims(self, q):
        return self._kernels[q].active_dims

    def filter_non_indep_idxs(self, idxs):
        lim = self._num_lmc + self._num_slfm
        return [q for q in idxs if q < lim]

    # -- evaluation on distances (host, O(m) per kernel) -------------------------
    def eval_kernels(self, dists):
        assert self.P
        return [k.from_dist(dists[k.active_dims]) for k in self._kernels]

    def eval_kernels_fixed_dim(self, dists, active_dim):
        return np.array([self._kernels[q].from_dist(dists)
                         for q in self.active_dims[active_dim]])

    def eval_kernel_gradients(self, dists):
        assert self.P
        return [k.kernel_gradient(dists[k.active_dims]) for k in self._kernels]

    # -- parameters -----------------------------------------------------------
    @property
    def noise(self):
        return self._noise

    @noise.setter
    def noise(self, value):
        self._noise[:] = value

    @property
    def coreg_vecs(self):
        return self._coreg_vecs

    @coreg_vecs.setter
    def coreg_vecs(self, values):
        for mine, theirs in zip(self._coreg_vecs, values):
            mine[:] = theirs

    @property
    def coreg_diags(self):
        return self._coreg_diags

    @coreg_diags.setter
    def coreg_diags(self, values):
        for mine, theirs in zip(self._coreg_diags, values):
            mine[:] = theirs

    def coreg_mats(self, active_dim=None):
        idx = (range(self.Q) if active_dim is None
               else self.active_dims[active_dim])
        return [self._coreg_vecs[q].T.dot(self._coreg_vecs[q]) +
                np.diag(self._coreg_diags[q]) for q in idx]

    # -- gradient sink ----------------------------------------------------------
    def update_gradient(self, grads):
        """Pull the four gradient families out of a likelihood object
        (reference functional_kernel.py:212-223)."""
        assert self.P
        for slot, g in zip(self.coreg_vec_grads, grads.coreg_vec_gradients()):
            slot[:] = g
        for slot, g in zip(self.coreg_diag_grads, grads.coreg_diags_gradients()):
            slot[:] = g
        for k, g in zip(self._kernels, grads.kernel_gradients()):
            k.update_gradient(g)
        self.noise_grad[:] = grads.noise_gradient()
