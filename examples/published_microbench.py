#!/usr/bin/env python3
"""The reference's PUBLISHED micro-benchmark configurations on one MI355X, next
to the numbers the reference published for them (its own CPU runs, in
/root/reference/benchmarks/.../out; restated below so that this file needs no
reference tree).

  inversion (benchmarks/representation-cmp/out/inv-run-1.txt:1-20; bench.py
  `inv`: one solve K alpha = y to the reference's rule, tol 1e-4, n = 5000,
  kernel family mix, eps 0.1, seed 1234):
      n_o 2500 d 2  r_q 2  q 10   best representation bt   0.8025 s  500 iterations
      n_o 500  d 10 r_q 1  q 10   best representation slfm 2.6700 s 1300 iterations
      n_o 500  d 10 r_q 10 q 1    best representation sum  0.2670 s  200 iterations
  gradient step (benchmarks/grad-grid/out/n5000-d10-r3-q1-eps0.01-krbf-run0.txt:13-43;
  bench.py `opt`: n_o 500, d 10, r_q 3, q 1, eps 0.01, rbf, seed 12340, 10 probes):
      solve K alpha = y + 10 trace terms 2.5625 s; 51 partial derivatives 0.9852 s;
      one optimisation iteration 3.5477 s (dense Cholesky path: 100.84 s)

Inputs follow the reference's generator (benchmarks/benchlib/bench.py:105-140,
restated in runlmc_amd/util/synth.py).  The device operator is ONE form for
every (D, Q, R) -- the reference's three representations are the same linear
map --, so there is one GPU time per configuration.

    python examples/published_microbench.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd.util import synth                                   # noqa: E402
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel              # noqa: E402
from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood           # noqa: E402
from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService  # noqa: E402
from runlmc_amd.approx.iterative import Iterative                   # noqa: E402

INV = [  # (n_o, d, r_q, q, published best representation, seconds, iterations)
    (2500, 2, 2, 10, 'bt', 0.8025, 500),
    (500, 10, 1, 10, 'slfm', 2.6700, 1300),
    (500, 10, 10, 1, 'sum', 0.2670, 200),
]


def best_of(fn, reps=3):
    out, best = None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    return out, best


def main():
    print('%-34s %12s %10s | %12s %10s %12s' % ('configuration', 'published s', 'iterations',
                                                'MI355X s', 'iterations', 'residual'))
    # Two modes.  'rule': MINRES's own stopping tests off (RL_MINRES_RULE), so a solve ends
    # on the reference's residual rule (iterative.py:36-42) -- how the PUBLISHED runs ended
    # (counts are multiples of 100, residuals 5e-6 ... 1e-4).  'scipy 1.15 exits': what
    # Iterative.solve does on today's SciPy -- minres's test1 <= rtol exit fires first.
    for mode, exits in (('rule', False), ('scipy 1.15 exits', True)):
        print('-- %s' % mode)
        for n_o, d, r, q, rep, pub_s, pub_it in INV:
            p = synth.make_problem(d, q, r, n_o, eps=0.1, seed=1234, kern='mix')
            fk = synth.functional_kernel(p)
            K, _ = gen_grid_kernel(fk, {(0,): p.grid_dists}, {(0,): (p.W, p.WT)}, p.lens)
            (x, it, err), sec = best_of(lambda: Iterative.solve(K, p.y, verbose=True, tol=1e-4,
                                                                scipy_exits=exits))
            print('%-34s %12.4f %10d | %12.4f %10d %12.3e'
                  % ('inv n_o %d d %d r_q %d q %d (%s)' % (n_o, d, r, q, rep), pub_s, pub_it, sec,
                     it, err))
    # the gradient step
    n_o, d, r, q, eps, seed, n_it = 500, 10, 3, 1, 0.01, 12340, 10
    p = synth.make_problem(d, q, r, n_o, eps=eps, seed=seed, kern='rbf')
    fk = synth.functional_kernel(p)
    ad = (0,)
    svc = StochasticDerivService(None, None, n_it, 1e-4)
    np.random.seed(1)
    probes = np.random.randint(0, 2, (n_it, p.n)) * 2 - 1

    def step():
        K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys, svc,
                                  probes=probes)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(), lik.kernel_gradients(),
             lik.noise_gradient())
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        nparam = sum(np.size(x) for x in g[0] + g[1]) + sum(len(x) for x in g[2]) + np.size(g[3])
        return t1 - t0, t2 - t1, nparam, float(np.mean(lik.deriv.iterations))

    print('\nopt n_o 500 d 10 r_q 3 q 1 eps 0.01 rbf: published 2.5625 s (solve alpha + 10 trace '
          'terms) + 0.9852 s (51 partial derivatives) = 3.5477 s per optimisation iteration')
    for mode, exits in (('rule', False), ('scipy 1.15 exits', True)):
        Iterative.SCIPY_EXITS = exits
        best = None
        for _ in range(3):
            cur = step()
            if best is None or cur[0] + cur[1] < best[0] + best[1]:
                best = cur
        print('MI355X (%s): %.4f s (solve alpha + 10 trace terms, %.0f iterations mean) + %.4f s '
              '(%d partial derivatives) = %.4f s' % (mode, best[0], best[3], best[1], best[2],
                                                     best[0] + best[1]))
    Iterative.SCIPY_EXITS = True


if __name__ == '__main__':
    main()
