"""Accuracy of the Hutchinson gradient against the EXACT-trace gradient, as the
reference's `bench.py opt` reports it (benchmarks/benchlib/bench.py:272-280:
err:grad l1 / l2 ratios), at N = 16 and N = 128 probes.

Exact side (oracle, test infrastructure): the reference's own per-parameter
loops fed with the complete basis as "probes" (r_i = sqrt(n) e_i, K~^-1 r_i from
a dense Cholesky of the stored SKI matrix), which turns (1/N) sum_i r_i^T K^-1 dK r_i
into tr(K^-1 dK) exactly.  Device side: fresh Rademacher probes, batched MINRES
solves, batched Gram gradients.  Runs on the small golden models (n <= 165); at C2 / C5 no exact trace is computable, the sampling error there
follows the same 1/sqrt(N) law.

    python tools/grad_accuracy.py            # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import scipy.linalg as la

from cases import Case
import parity_suite as ps
from oracle import likelihood as olik
from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService


def flat(g):
    return np.concatenate([np.ravel(x) for x in g['coreg_vec']] + [np.ravel(x) for x in g['coreg_diag']] +
                          [np.ravel(np.array(x)) for x in g['kernel']] + [np.ravel(g['noise'])])


def main():
    print('%-10s %6s %5s %12s %12s %12s' % ('case', 'n', 'N', 'err:grad l1', 'err:grad l2', 'alpha l2 err'))
    for name in ('lmc_small', 'lmc_c1', 'lmc_q1', 'lmc_2d'):
        c = Case(name)
        n = c.n
        if 'K_dense' in c.g:
            Kd = c.g['K_dense']
        else:       # dense SKI matrix from the oracle's operator, column by column
            Kd = olik.LMCOperatorOracle(c.spec(), c.grid_dists, c.W, c.WT, c.lens,
                                        active_dim=c.ad).as_numpy()
            Kd = 0.5 * (Kd + Kd.T)
        cf = la.cho_factor(Kd)
        alpha = la.cho_solve(cf, c.y)
        basis = np.sqrt(n) * np.identity(n)
        inv = np.sqrt(n) * la.cho_solve(cf, np.identity(n))
        exact = flat(olik.stochastic_gradients(c.spec(), c.grid_dists, c.W, c.WT, c.lens, alpha,
                                               basis, inv, active_dim=c.ad))
        fk, K, gk = ps.build_operator(c)
        for N in (16, 128):
            errs1, errs2, aerr = [], [], []
            for rep in range(5):
                rng = np.random.RandomState(1000 * N + rep)
                rs = rng.randint(0, 2, (N, n)) * 2 - 1
                svc = StochasticDerivService(None, None, N, 1e-4)
                lik = ApproxLMCLikelihood(fk, K, {c.ad: c.grid_dists}, {c.ad: (c.W, c.WT)},
                                          c.Ys, svc, probes=rs)
                got = flat(dict(coreg_vec=lik.coreg_vec_gradients(),
                                coreg_diag=lik.coreg_diags_gradients(),
                                kernel=lik.kernel_gradients(), noise=lik.noise_gradient()))
                e = got - exact
                errs1.append(np.abs(e).sum() / np.abs(exact).sum())
                errs2.append(np.linalg.norm(e) / np.linalg.norm(exact))
                aerr.append(np.linalg.norm(lik.alpha() - alpha) / np.linalg.norm(alpha))
            print('%-10s %6d %5d %12.3e %12.3e %12.3e' % (name, n, N, np.mean(errs1), np.mean(errs2),
                                                        np.mean(aerr)), flush=True)


if __name__ == '__main__':
    main()
