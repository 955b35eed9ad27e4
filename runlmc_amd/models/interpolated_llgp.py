"""The caller of the hot path: an InterpolatedLLGP-compatible model without
paramz (mirror of reference runlmc/models/interpolated_llgp.py:24-443 and
multigp.py:18-200).

Same constructor arguments, ``parameters_changed()`` / ``optimize()`` /
``predict()`` / ``log_likelihood()`` / ``normal_quadratic()`` / ``log_det_K()``.
Every product, solve and gradient inside runs on the device
(runlmc_amd.lmc.*).  Differences, all stated:

* ``log_det_K()`` is the matrix-free stochastic-Lanczos estimate of
  log det K~ (the reference's is a dense Cholesky of the exact kernel and is
  never on its optimiser path, interpolated_llgp.py:262-276).
* inputs of any dimension, each kernel acting on one or two of them (bicubic
  interpolation, BTTB kernels; kernels on different active-dimension sets get
  their own grids);
* prediction modes: 'on-the-fly' and 'precompute' (both batched solves on the
  device); 'exact' (dense Cholesky) is not provided.
* parameters live in one flat array in the optimiser's space; positive
  parameters go through paramz's Logexp (softplus) transform as in the
  reference (functional_kernel.py:130,185; rbf.py:33).
* `max_procs` is accepted and ignored (no process pool).
"""
import logging

import numpy as np
import scipy.spatial.distance as sdist

from ..approx.interpolation import autogrid, multi_interpolant
from ..approx.iterative import Iterative
from ..lmc.grid_kernel import gen_grid_kernel
from ..lmc.likelihood import ApproxLMCLikelihood
from ..lmc.metrics import Metrics
from ..lmc.stochastic_deriv import StochasticDerivService
from .optimization import AdaDelta

_LOG = logging.getLogger(__name__)
_LIM = 36.0       # paramz Logexp switches to the identity beyond this


def _softplus(x):
    x = np.asarray(x, dtype=float)
    return np.where(x > _LIM, x, np.log1p(np.exp(np.clip(x, -np.inf, _LIM))))


def _softplus_inv(f):
    f = np.asarray(f, dtype=float)
    return np.where(f > _LIM, f, np.log(np.expm1(np.clip(f, 1e-300, _LIM))))


def _softplus_grad(f):
    """d softplus / dx expressed through the value f (paramz Logexp.gradfactor)."""
    f = np.asarray(f, dtype=float)
    return np.where(f > _LIM, 1.0, -np.expm1(-f))


class InterpolatedLLGP:
    EVAL_NORM = np.inf

    def __init__(self, Xs, Ys, normalize=True, lo=None, hi=None, m=None,
                 name='lmc', metrics=False, prediction='on-the-fly',
                 max_procs=None, trace_iterations=15, tolerance=1e-4,
                 functional_kernel=None, group=None, device_index=0, device_probes=None):
        self.name = name
        self.input_dim, self.output_dim = self._validate_io(Xs, Ys)
        self.normalizer = None
        Ys = [np.asarray(Y, dtype=float) for Y in Ys]
        if normalize:
            self.normalizer = [(Y.mean(), Y.std()) for Y in Ys]
            if any(s == 0 for _, s in self.normalizer):
                raise ValueError('an output has no variance')
            Ys = [(Y - mu) / sd for Y, (mu, sd) in zip(Ys, self.normalizer)]
        self.Ys = Ys
        self.Xs = [np.asarray(X, dtype=float).reshape(len(X), -1) for X in Xs]
        if not functional_kernel:
            raise ValueError('functional_kernel must be provided')
        if prediction not in ('on-the-fly', 'precompute'):
            raise ValueError('Variance prediction method {} unrecognized'
                             .format(prediction))
        self.prediction = prediction
        self._functional_kernel = functional_kernel
        self._functional_kernel.set_input_dim(self.input_dim)
        if any(len(ad) > 2 for ad in functional_kernel.active_dims):
            raise NotImplementedError(
                'kernels may act on one or two input dimensions each')
        self.y = np.hstack(self.Ys)
        self.kernel = None
        self.dists, self.interpolants, self.grid_axes = {}, {}, {}
        self._generate_grids(lo, hi, m)
        self.metrics = Metrics() if metrics else None
        self._device_index = device_index
        # (device_probes: a seed -- the Hutchinson probes are drawn on the device instead of by
        # NumPy's global RNG, StochasticDerivService; None keeps the reference's stream)
        self._deriv_service = StochasticDerivService(
            self.metrics, None, trace_iterations, tolerance, group=group,
            device_probes=device_probes)
        self._K = None
        self._grid_kernels = None
        self._caches = {}
        _LOG.info('InterpolatedLLGP %s fully initialized', self.name)

    # -- data handling (multigp.py:75-118) ----------------------------------------
    @staticmethod
    def _validate_io(Xs, Ys):
        if len(Xs) != len(Ys):
            raise ValueError('Xs and Ys must have one entry per output')
        if not Xs:
            raise ValueError('need at least one output')
        dims = set()
        for X, Y in zip(Xs, Ys):
            X = np.asarray(X)
            if len(X) != len(Y):
                raise ValueError('an X and its Y differ in length')
            dims.add(1 if X.ndim == 1 else X.shape[1])
        if len(dims) != 1:
            raise ValueError('inputs differ in dimension')
        return dims.pop(), len(Ys)

    @staticmethod
    def _wrap(v, active_dims):
        """Entries of lo / hi / m that belong to one active-dimension set
        (interpolated_llgp.py:406-413): a scalar only for a single active
        dimension, otherwise indexed by the set."""
        if v is None:
            return None
        a = np.asarray(v, dtype=float)
        if not a.shape:
            if len(active_dims) != 1:
                raise ValueError('scalar lo / hi / m needs a single active dimension, got %d'
                                 % len(active_dims))
            return a.reshape(1)
        return a[list(active_dims)]

    def _generate_grids(self, lo, hi, m):
        for ad in self._functional_kernel.active_dims:
            Xs = [X[:, list(ad)] for X in self.Xs]
            wlo, whi, wm = (self._wrap(v, ad) for v in (lo, hi, m))
            self.grid_axes[ad] = autogrid(Xs, wlo, whi, wm)
            axes = self.grid_axes[ad]
            # distance of every grid point to grid point 0, shaped like the
            # grid (interpolated_llgp.py:425-432)
            mesh = np.meshgrid(*[a - a[0] for a in axes], indexing='ij')
            self.dists[ad] = np.sqrt(sum(np.square(g) for g in mesh))
            W = multi_interpolant(Xs, *self.grid_axes[ad])
            WT = W.transpose().tocsr()
            WT.sort_indices()
            WT.indices = WT.indices.astype(np.int32)
            WT.indptr = WT.indptr.astype(np.int32)
            self.interpolants[ad] = (W, WT)

    # -- flat parameter vector in the optimiser's space ------------------------------
    def _param_blocks(self):
        """(array view, is_positive, gradient getter) for every free block, in
        a fixed order: coregionalisation vectors of LMC and SLFM kernels,
        kappa of LMC kernels, kernel parameters, noise."""
        fk = self._functional_kernel
        n_lmc, n_slfm = fk._num_lmc, fk._num_slfm
        blocks = []
        for q in range(n_lmc + n_slfm):
            blocks.append(('a%d' % q, fk.coreg_vecs[q], False,
                           lambda q=q: fk.coreg_vec_grads[q]))
        for q in range(n_lmc):
            blocks.append(('kappa%d' % q, fk.coreg_diags[q], True,
                           lambda q=q: fk.coreg_diag_grads[q]))
        for q, k in enumerate(fk.kernels):
            blocks.append(('kern%d' % q, k, True, lambda k=k: k.gradient))
        blocks.append(('noise', fk.noise, True, lambda: fk.noise_grad))
        return blocks

    @property
    def param_array(self):
        out = []
        for _, holder, positive, _ in self._param_blocks():
            vals = holder.param_array if hasattr(holder, 'set_params') else np.ravel(holder)
            out.append(_softplus_inv(vals) if positive else np.array(vals, dtype=float))
        return np.concatenate(out)

    @param_array.setter
    def param_array(self, x):
        x = np.asarray(x, dtype=float)
        pos = 0
        for _, holder, positive, _ in self._param_blocks():
            size = (len(holder.param_array) if hasattr(holder, 'set_params')
                    else holder.size)
            vals = x[pos:pos + size]
            pos += size
            vals = _softplus(vals) if positive else vals
            if hasattr(holder, 'set_params'):
                holder.set_params(vals)
            else:
                holder[...] = vals.reshape(holder.shape)
        assert pos == len(x)
        self.parameters_changed()

    @property
    def gradient(self):
        """d log-likelihood / d param_array (optimiser space)."""
        out = []
        for _, holder, positive, getter in self._param_blocks():
            g = np.ravel(np.asarray(getter(), dtype=float))
            if positive:
                vals = holder.param_array if hasattr(holder, 'set_params') else np.ravel(holder)
                g = g * _softplus_grad(vals)
            out.append(g)
        return np.concatenate(out)

    # -- the step (interpolated_llgp.py:192-245) -------------------------------------
    def parameters_changed(self):
        self._caches.clear()
        fk = self._functional_kernel
        lens = [len(Y) for Y in self.Ys]
        if self._K is None:
            self._K, self._grid_kernels = gen_grid_kernel(
                fk, self.dists, self.interpolants, lens,
                device_index=self._device_index)
        else:
            # same grid and interpolants: only spectra, factors and noise change
            for ad, gk in self._grid_kernels.items():
                gk.update(fk, self.dists[ad])
            self._K.update_noise(fk.noise, lens)
        self.kernel = ApproxLMCLikelihood(
            fk, self._K, self.dists, self.interpolants, self.Ys,
            self._deriv_service)
        self.kernel._grid_kernels = self._grid_kernels
        fk.update_gradient(self.kernel)
        if self.metrics is not None:
            g = self.gradient
            self.metrics.grad_norms.append(float(np.abs(g).max()))
            self.metrics.log_likely.append(self.log_likelihood())

    def optimize(self, optimizer=None, **kwargs):
        """Maximise the likelihood with AdaDelta (reference multigp.py:176-197
        through paramz)."""
        if self.metrics is not None:
            self.metrics = Metrics()
            self._deriv_service.metrics = self.metrics
        opt = optimizer or AdaDelta(**kwargs)
        x = self.param_array.copy()

        def neg_grad(xx):
            self.param_array = xx
            return -self.gradient

        try:
            opt.opt(x, neg_grad)
        except KeyboardInterrupt:
            _LOG.warning('optimization interrupted; keeping current parameters')
            raise
        self.param_array = x
        return opt

    # -- likelihood terms -----------------------------------------------------------
    def _ensure(self):
        if self.kernel is None:
            self.parameters_changed()

    def normal_quadratic(self):
        self._ensure()
        return self.kernel.normal_quadratic()

    def log_det_K(self):
        self._ensure()
        return self.kernel.log_det_K()

    def log_likelihood(self):
        self._ensure()
        return self.kernel.log_likelihood()

    # -- prediction (interpolated_llgp.py:293-397) --------------------------------------
    def _grid_alpha(self):
        if 'grid_alpha' not in self._caches:
            out = {}
            for ad, gk in self._grid_kernels.items():
                _, WT = self.interpolants[ad]
                out[ad] = gk.grid_K.matvec(WT.dot(self.kernel.alpha()))
            self._caches['grid_alpha'] = out
        return self._caches['grid_alpha']

    def _native_variance(self):
        fk = self._functional_kernel
        coregs = np.column_stack([np.square(a).sum(axis=0) for a in fk.coreg_vecs])
        coregs = coregs + np.column_stack(fk.coreg_diags)
        zero = {ad: 0 for ad in fk.active_dims}
        k0 = np.array(fk.eval_kernels(zero), dtype=float)
        return coregs.dot(k0).reshape(-1) + fk.noise

    def _exact_cross_kernel(self, Xs):
        """Exact (non-SKI) covariance between test and training points
        (reference likelihood.py:176-199)."""
        fk = self._functional_kernel
        rl, cl = [len(X) for X in Xs], [len(X) for X in self.Xs]
        A = np.vstack([X.reshape(len(X), self.input_dim) for X in Xs])
        B = np.vstack(self.Xs)
        dist = {ad: sdist.cdist(A[:, list(ad)], B[:, list(ad)])
                for ad in fk.active_dims}
        ro, co = np.repeat(np.arange(fk.D), rl), np.repeat(np.arange(fk.D), cl)
        K = np.zeros((sum(rl), sum(cl)))
        for Bq, k in zip(fk.coreg_mats(), fk.kernels):
            K += Bq[np.ix_(ro, co)] * k.from_dist(dist[k.active_dims])
        return K

    def _var_on_the_fly(self, _W, Xs):
        Kx = self._exact_cross_kernel(Xs)
        if Kx.shape[0] == 0:
            return np.zeros(0)
        sol = Iterative.solve(self._K, Kx)           # one batched device solve
        return np.einsum('ij,ij->i', Kx, np.atleast_2d(sol))

    def _precomputed_nu(self):
        if 'nu' not in self._caches:
            if len(self.interpolants) != 1:
                raise ValueError(
                    'precompute prediction mode unavailable for split kernels')
            (ad,) = self.interpolants
            W, WT = self.interpolants[ad]
            gk = self._grid_kernels[ad]
            Dm = W.shape[1]
            # K_XU e_i for every grid index: columns of W K_UU
            KXU = W.dot(gk.grid_K.matmat(np.identity(Dm)))          # (n, Dm)
            sol = Iterative.solve(self._K, np.ascontiguousarray(KXU.T))   # (Dm, n)
            back = gk.grid_K.matmat(WT.dot(np.atleast_2d(sol).T))    # K_UX K^-1 K_XU
            self._caches['nu'] = np.diag(back).copy()
        return self._caches['nu']

    def _var_precompute(self, Ws, _Xs):
        nu = self._precomputed_nu()
        (W,) = Ws.values()
        return W.dot(nu)

    def _raw_predict(self, Xs):
        self._ensure()
        Xs = [np.asarray(X, dtype=float).reshape(len(X), self.input_dim) for X in Xs]
        lens = [len(X) for X in Xs]
        mean = np.zeros(sum(lens))
        Ws = {}
        for ad, grid_alpha in self._grid_alpha().items():
            Ws[ad] = multi_interpolant([X[:, list(ad)] for X in Xs], *self.grid_axes[ad])
            mean += Ws[ad].dot(grid_alpha)
        native = np.repeat(self._native_variance(), lens)
        explained = (self._var_on_the_fly if self.prediction == 'on-the-fly'
                     else self._var_precompute)(Ws, Xs)
        var = native - explained
        var[var < 0] = 0
        cuts = np.cumsum(lens)[:-1]
        return np.split(mean, cuts), np.split(var, cuts)

    def predict(self, Xs):
        """(means, variances), one array per output, de-normalised
        (multigp.py:108-150)."""
        if len(Xs) != self.output_dim:
            raise ValueError('need one (possibly empty) input array per output')
        mu, var = self._raw_predict(Xs)
        if self.normalizer:
            mu = [m * sd + mean for m, (mean, sd) in zip(mu, self.normalizer)]
            var = [v * sd ** 2 for v, (_, sd) in zip(var, self.normalizer)]
        return mu, var

    def predict_quantiles(self, Xs, quantiles=(2.5, 97.5)):
        """Gaussian predictive quantiles (multigp.py:152-174)."""
        from scipy.stats import norm
        mu, var = self.predict(Xs)
        return [[m + norm.ppf(q / 100.0) * np.sqrt(v) for q in quantiles]
                for m, v in zip(mu, var)]
