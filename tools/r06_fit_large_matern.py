"""Round 6: a FIT (AdaDelta steps, parameters moving every step) of a large problem whose kernels are
all Matern-3/2 -- the operators whose solves run conjugate gradients preconditioned on the larger basis
(csrc/rl_solve.hip hz_*): every step rebuilds C_q and the map, the basis and the table stay.
    python tools/r06_fit_large_matern.py [n_per_output] [steps] [kern: matern|mix]
Prints per-step wall time, the solver's iteration counts / residuals of the step's solves and the
factorisation mode after each step."""
import os, sys, time
os.environ.setdefault('OMP_NUM_THREADS', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.kern.stationary import RBF, Matern32
from runlmc_amd.lmc.functional_kernel import FunctionalKernel
from runlmc_amd.models.interpolated_llgp import InterpolatedLLGP
from runlmc_amd.models.optimization import AdaDelta

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kern = sys.argv[3] if len(sys.argv) > 3 else 'matern'
D = 3
rng = np.random.RandomState(7)
xss = [np.sort(rng.rand(n1)) for _ in range(D)]
lat = lambda x: np.sin(9 * x) + 0.5 * np.sin(31 * x + 1) + 0.2 * np.abs(np.sin(57 * x))   # (a kink: Matern-like)
yss = [c * lat(x) + 0.3 * np.cos(13 * x + d) + 0.1 * rng.randn(n1) for d, (x, c) in enumerate(zip(xss, (1.0, -0.7, 0.4)))]
ks = [Matern32(inv_lengthscale=3.0, name='m0'), Matern32(inv_lengthscale=20.0, name='m1')] if kern == 'matern' else \
     [RBF(inv_lengthscale=3.0, name='r0'), Matern32(inv_lengthscale=20.0, name='m1')]
fk = FunctionalKernel(D=D, lmc_kernels=ks, lmc_ranks=[1, 1])
t0 = time.perf_counter()
lmc = InterpolatedLLGP(xss, yss, functional_kernel=fk, normalize=True, m=n1 // 5)
print('model: n = %d, grid %s, set-up %.2f s' % (D * n1, [len(a) for ax in lmc.grid_axes.values() for a in ax], time.perf_counter() - t0), flush=True)
stamps = [time.perf_counter()]

def cb():
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())
    line = 'step %2d: %.3f s' % (len(stamps) - 1, stamps[-1] - stamps[-2])
    try:
        lik = lmc.kernel
        d = lik.deriv
        line += ' | solves: iterations max %d, residual max %.2e | mode %s' % (
            int(np.max(d.iterations)), float(np.max(d.residuals)),
            lik.K.device_operator().factor_mode if hasattr(lik, 'K') else '?')
    except Exception as e:                                        # the attribute names are the model's business
        line += ' | (%s)' % type(e).__name__
    print(line, flush=True)

opt = AdaDelta(max_it=steps, callback=cb, permitted_drops=10 ** 6)
lmc.optimize(optimizer=opt)
try:
    nll = -lmc.log_likelihood()
except ValueError:                     # (preconditioned solves: no log det on that path)
    nll = float('nan')
print('fit: %d steps, %.2f s; NLL %s; inv length scales %s; noise %s' % (
    opt.n_iter, stamps[-1] - stamps[0], nll,
    [float(np.ravel(getattr(k, 'inv_lengthscale', np.nan))[0]) for k in ks], np.round(fk.noise, 4)))
