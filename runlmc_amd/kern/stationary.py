"""Stationary kernels evaluated on grid distances (host NumPy, O(m)).

Paramz-free counterparts of reference runlmc/kern/{rbf,matern32,std_periodic,
scaled,stationary_kern}.py: only ``from_dist``, ``kernel_gradient``,
``update_gradient`` and ``active_dims`` are on the hot path's input side."""
import numpy as np


class StationaryKern:
    """A kernel k(r) of distance with differentiable parameters."""

    def __init__(self, name, active_dims=None):
        self.name = name
        self.active_dims = active_dims
        self.gradient = None

    def from_dist(self, dists):
        raise NotImplementedError

    def kernel_gradient(self, dists):
        """List of dk/dtheta_p arrays, one per parameter."""
        raise NotImplementedError

    def update_gradient(self, grad):
        self.gradient = np.asarray(grad, dtype=float)

    @property
    def param_array(self):
        raise NotImplementedError


class RBF(StationaryKern):
    """exp(-gamma r^2 / 2) (reference rbf.py:39-54)."""

    def __init__(self, inv_lengthscale=1, name='rbf', active_dims=None):
        super().__init__(name, active_dims)
        self.inv_lengthscale = float(inv_lengthscale)

    def from_dist(self, dists):
        return np.exp(-0.5 * np.square(dists) * self.inv_lengthscale)

    def kernel_gradient(self, dists):
        sq = np.square(dists)
        return [np.exp(-0.5 * sq * self.inv_lengthscale) * (-0.5 * sq)]

    @property
    def param_array(self):
        return np.array([self.inv_lengthscale])

    def set_params(self, p):
        self.inv_lengthscale = float(p[0])


class Matern32(StationaryKern):
    """(1 + s) exp(-s), s = sqrt(3) gamma r (reference matern32.py:39-57)."""

    def __init__(self, inv_lengthscale=1, name='matern32', active_dims=None):
        super().__init__(name, active_dims)
        self.inv_lengthscale = float(inv_lengthscale)

    def from_dist(self, dists):
        s = dists * np.sqrt(3) * self.inv_lengthscale
        return (1 + s) * np.exp(-s)

    def kernel_gradient(self, dists):
        root3r = dists * np.sqrt(3)
        s = root3r * self.inv_lengthscale
        e = np.exp(-s)
        return [(1 + s) * (-root3r * e) + root3r * e]

    @property
    def param_array(self):
        return np.array([self.inv_lengthscale])

    def set_params(self, p):
        self.inv_lengthscale = float(p[0])


class StdPeriodic(StationaryKern):
    """exp(-gamma sin^2(pi r / T) / 2) (reference std_periodic.py:44-67)."""

    def __init__(self, inv_lengthscale=1, period=1, name='std_periodic',
                 active_dims=None):
        super().__init__(name, active_dims)
        self.inv_lengthscale = float(inv_lengthscale)
        self.period = float(period)

    def from_dist(self, dists):
        if np.log(self.period) < -200:
            return np.nan
        s = np.sin((np.pi / self.period) * dists)
        return np.exp(-0.5 * np.square(s) * self.inv_lengthscale)

    def kernel_gradient(self, dists):
        arg = np.pi / self.period * dists
        s = np.sin(arg)
        ds = np.cos(arg) * arg * (-1 / self.period * self.inv_lengthscale)
        sq = np.square(s)
        e = np.exp(-0.5 * sq * self.inv_lengthscale)
        return [e * (-0.5 * sq), e * (-1 * s * ds)]

    @property
    def param_array(self):
        return np.array([self.inv_lengthscale, self.period])

    def set_params(self, p):
        self.inv_lengthscale, self.period = float(p[0]), float(p[1])


class Scaled(StationaryKern):
    """scale * k(r); the scale is the LAST parameter (reference
    runlmc/kern/scaled.py:13-37)."""

    def __init__(self, k, scale=1.0):
        super().__init__('scaled_' + k.name, k.active_dims)
        self.k = k
        self.scale = float(scale)

    def from_dist(self, dists):
        return self.scale * self.k.from_dist(dists)

    def kernel_gradient(self, dists):
        return ([self.scale * g for g in self.k.kernel_gradient(dists)] +
                [self.k.from_dist(dists)])

    def update_gradient(self, grad):
        # the reference creates `scale` but never links it (scaled.py:20), so
        # it is not optimised: the free parameters are the inner kernel's
        grad = np.asarray(grad, dtype=float)
        self.gradient = grad[:-1]
        self.scale_gradient = float(grad[-1])
        self.k.update_gradient(grad[:-1])

    @property
    def param_array(self):
        return self.k.param_array

    def set_params(self, p):
        self.k.set_params(p)
