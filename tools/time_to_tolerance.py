"""Solves run to the REFERENCE's tolerance (approx/iterative.py:36-58: explicit
residual < 1e-4 at every 100th iteration) with MINRES's own stopping tests off
(RL_MINRES_RULE / oracle own_exits=False), next to the default mode in which
SciPy 1.15's test1 exit ends them early.

    python tools/time_to_tolerance.py [c2|c5] [kern] [maxiter] [oracle_systems]
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
from runlmc_amd._native import solve_batch, MINRES, MINRES_RULE     # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    kern = sys.argv[2] if len(sys.argv) > 2 else 'rbf'
    maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n_oracle = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    D, Q, R, m, npr = synth.CONFIGS[cfg]
    p = synth.make_problem(D, Q, R, m, kern=kern)
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    op = K.device_operator()
    np.random.seed(4321)
    rs = np.random.randint(0, 2, (npr, p.n)) * 2.0 - 1
    B = np.vstack([p.y, rs])
    Bd = torch.from_numpy(B).to(op.device)
    print('%s %s: n = %d, %d systems, tol 1e-4, check every 100, maxiter %s'
          % (cfg, kern, p.n, len(B), maxiter or 'n'))
    out = {}
    for name, method in (('scipy-1.15 exits', MINRES), ('reference rule only', MINRES_RULE)):
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            X, it, res, st = solve_batch(op, Bd, method=method, tol=1e-4, maxiter=maxiter)[:4]
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        it, res = np.asarray(it), np.asarray(res)
        out[name] = (X, it, res)
        print('  %-20s %8.3f s | iterations min %d mean %.0f max %d | residual max %.3e median %.3e '
              '| systems below 1e-4: %d of %d | exit codes %s'
              % (name, best, it.min(), it.mean(), it.max(), res.max(), np.median(res),
                 int((res < 1e-4).sum()), len(res), sorted(set(int(s) for s in st))), flush=True)
    if n_oracle:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from oracle import likelihood as olik
        from oracle.kernels import KernelSpec, RBFSpec, Matern32Spec, StdPeriodicSpec
        from oracle.solver import iterative_solve
        make = {'rbf': RBFSpec, 'periodic': StdPeriodicSpec, 'matern': Matern32Spec}
        spec = KernelSpec(p.D, [make[d[0]](*d[1:]) for d in p.kern_desc], list(p.coreg_vecs),
                          list(p.coreg_diags), p.noise)
        spec.set_input_dim(1)
        oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
        X, it, res = out['reference rule only']
        for v in range(n_oracle):
            t0 = time.perf_counter()
            xo, ito, erro, ok = iterative_solve(oop.matvec, B[v], tol=1e-4, own_exits=False)
            el = time.perf_counter() - t0
            xd = X[v].cpu().numpy()
            print('  oracle system %d (one core): %.1f s, %d iterations, residual %.3e | device %d, '
                  '%.3e | iterates differ by %.2e of max|x|'
                  % (v, el, ito, erro, it[v], res[v], np.abs(xd - xo).max() / np.abs(xo).max()),
                  flush=True)


if __name__ == '__main__':
    main()
