"""AdaDelta driver with the reference's gradient-norm stopping rule
(paramz/climin-free mirror of reference runlmc/models/optimization.py:13-83).

The update is climin's Adadelta with momentum (what the reference wraps):

    d      = momentum * step_prev;  x -= d
    g      = grad(x)
    gms    = decay * gms + (1 - decay) * g^2
    step2  = sqrt(sms + offset) / sqrt(gms + offset) * g * step_rate
    x     -= step2;  step = d + step2
    sms    = decay * sms + (1 - decay) * step^2

Stopping (optimization.py:66-80): track the rolling maximum of the gradient
infinity-norm; every iteration whose norm is below ``min_grad_ratio`` times
that maximum uses up one of ``permitted_drops``; stop when they run out or at
``max_it``.
"""
import numpy as np


class AdaDelta:
    def __init__(self, **kwargs):
        self.kwargs = {'step_rate': 1, 'decay': 0.9, 'momentum': 0.5,
                       'offset': 1e-4, 'max_it': 100, 'verbosity': 0,
                       'min_grad_ratio': 0.1, 'permitted_drops': 5,
                       'callback': lambda: None}
        self.kwargs.update(kwargs)
        self.x_opt = None
        self.n_iter = 0

    def opt(self, x, fp):
        """Minimise in place over the array `x`; fp(x) returns the gradient of
        the objective (the negative log likelihood)."""
        k = self.kwargs
        gms = np.zeros_like(x)
        sms = np.zeros_like(x)
        step = np.zeros_like(x)
        rolling_max, drops = 0.0, k['permitted_drops']
        if k['verbosity']:
            print('starting adadelta', {a: b for a, b in k.items() if a != 'callback'})
        n_iter = 0
        while True:
            n_iter += 1
            d = k['momentum'] * step
            x -= d
            g = fp(x)
            gms = k['decay'] * gms + (1 - k['decay']) * g ** 2
            step2 = np.sqrt(sms + k['offset']) / np.sqrt(gms + k['offset']) * g * k['step_rate']
            x -= step2
            step = d + step2
            sms = k['decay'] * sms + (1 - k['decay']) * step ** 2
            grad_norm = np.abs(g).max() if g.size else 0.0
            rolling_max = max(rolling_max, grad_norm)
            if k['verbosity']:
                every = max(k['max_it'] // k['verbosity'], 1)
                if n_iter % every == 0:
                    print('iteration {:8d} grad norm {:10.4e}'.format(n_iter, grad_norm))
            k['callback']()
            if grad_norm < k['min_grad_ratio'] * rolling_max:
                drops -= 1
            if n_iter >= k['max_it'] or drops <= 0:
                break
        self.n_iter = n_iter
        self.x_opt = x
        return x
