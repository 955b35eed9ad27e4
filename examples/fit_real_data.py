"""End-to-end fits of the reference's benchmark workloads on the GPU (BASELINE
configs 3 and 4 and the synthetic two-input benchmark): hyper-parameter
optimisation with AdaDelta and the reference's stopping rule, then held-out
prediction, scored with the reference's SMSE and NLPD
(benchmarks/benchlib/standard_tester.py:205-266).

    python examples/fit_real_data.py [fx2007|weather|weather1000|synth] [runs]

Data: tests/golden/fit_*.npz (train / held-out splits derived from the
reference's data files by tests/golden/make_golden.py --fit-data-only).
Published by the reference (16 CPU processes, BASELINE.md;
benchmarks/weather-out/results_weather.tex, paper/results_synth.tex):
    FX2007              69 s   SMSE 0.21   NLPD -3.62
    weather, m = 500    73 s   SMSE 0.09   NLPD  1.72
    weather, m = 1000   90 s   SMSE 0.09   NLPD  1.69
    synthetic 2-D      161 s   SMSE 0.12   NLPD  0.28   (D=5, n=47.5k, m=25x25, tol 1e-3)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from runlmc_amd.kern.stationary import RBF, Scaled                       # noqa: E402
from runlmc_amd.lmc.functional_kernel import FunctionalKernel           # noqa: E402
from runlmc_amd.models.interpolated_llgp import InterpolatedLLGP        # noqa: E402
from runlmc_amd.models.optimization import AdaDelta                     # noqa: E402


def load(name):
    name = 'weather' if name.startswith('weather') else name
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_%s.npz' % name))
    D = len(d['names'])
    get = lambda k: [d['%s%d' % (k, i)] for i in range(D)]               # noqa: E731
    return get('x'), get('y'), get('tx'), get('ty')


def smse(test_yss, pred_yss, train_yss):
    # standard_tester.py:205-211
    vals = [np.square(t - p).mean() / np.square(tr.mean() - t).mean()
            for t, p, tr in zip(test_yss, pred_yss, train_yss) if len(t)]
    return float(np.mean(vals))


def nlpd(test_yss, pred_yss, pred_vss):
    # standard_tester.py:214-233 (zero predictive variances are dropped there)
    vals = []
    for t, p, v in zip(test_yss, pred_yss, pred_vss):
        keep = np.flatnonzero(v)
        if len(keep):
            t, p, v = t[keep], p[keep], v[keep]
            vals.append(0.5 * np.mean(np.square(t - p) / v + np.log(2 * np.pi * v)))
    return float(np.mean(vals))


def kernel_for(name, D):
    if name == 'fx2007':
        # Alvarez and Lawrence: vanilla LMC, Q = 1, rank 2 (standard_tester.py:48-53)
        return FunctionalKernel(D=D, lmc_kernels=[RBF(name='rbf0')], lmc_ranks=[2]), None, \
            {'min_grad_ratio': 0.2}, {}
    if name == 'synth':
        # Q = 2 SLFM + one independent RBF per output on two inputs, 25 x 25
        # interpolating points, solver tolerance 1e-3 (benchmarks/synth/synth.py:30-55,
        # standard_tester.py:454-458)
        return (FunctionalKernel(D=D, slfm_kernels=[RBF(name='rbf1'), RBF(name='rbf2')],
                                 indep_gp=[RBF(name='indep%d' % i) for i in range(D)]),
                [25, 25], {}, {'tolerance': 1e-3})
    # rank-2 SLFM + one independent Scaled(RBF) per output (standard_tester.py:35-45)
    return (FunctionalKernel(D=D, slfm_kernels=[RBF(name='slfm%d' % i) for i in range(2)],
                             indep_gp=[Scaled(RBF(name='rbf%d' % i)) for i in range(D)]),
            1000 if name == 'weather1000' else 500, {}, {})


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'fx2007'
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    xss, yss, test_xss, test_yss = load(name)
    np.random.seed(1234)
    times, smses, nlpds = [], [], []
    for r in range(runs):
        fk, m, opt_opts, model_opts = kernel_for(name, len(xss))
        lmc = InterpolatedLLGP(xss, yss, functional_kernel=fk, normalize=True, m=m,
                               **model_opts)
        opt = AdaDelta(**opt_opts)
        t0 = time.perf_counter()
        lmc.optimize(optimizer=opt)
        times.append(time.perf_counter() - t0)
        P = 1 if np.ndim(xss[0]) == 1 else np.shape(xss[0])[1]
        pred_yss, pred_vss = lmc.predict([np.reshape(x, (len(x), P)) for x in test_xss])
        smses.append(smse(test_yss, pred_yss, yss))
        nlpds.append(nlpd(test_yss, pred_yss, pred_vss))
        print('%s run %d: %d AdaDelta steps, fit %.2f s, SMSE %.3f, NLPD %.3f' % (
            name, r, opt.n_iter, times[-1], smses[-1], nlpds[-1]), flush=True)
    se = lambda v: np.std(v) / np.sqrt(len(v))                           # noqa: E731
    grid = 'x'.join(str(len(a)) for axes in lmc.grid_axes.values() for a in axes)
    print('%s: n = %d, D = %d, grid %s | fit %.2f (%.2f) s | SMSE %.3f (%.3f) | NLPD %.3f (%.3f)'
          ' | medians: fit %.2f s, SMSE %.3f, NLPD %.3f' % (
              name, sum(map(len, xss)), len(xss), grid, np.mean(times), se(times),
              np.mean(smses), se(smses), np.mean(nlpds), se(nlpds), np.median(times),
              np.median(smses), np.median(nlpds)))


if __name__ == '__main__':
    main()
