"""Iteration parity report: solver iterations, final residuals and
wall-clock of the device solve against the oracle's statement of the
reference's Iterative.solve, on the C2 synthetic system at eps = 0.1 (the
reference's ill-conditioned default) and eps = 1, with the reference's check
period (100) and a per-iteration check.

A report, not a test (tests/ is the only place besides bench.py's cpu_baseline
that may import oracle/).  Run on the GPU box:
    python tests/report_iteration_parity.py [c2|c1|c5]
"""
import os
os.environ.setdefault('OMP_NUM_THREADS', '1')   # reference bench.py:7 (a BLAS thread pool of cpu_count() threads under a cgroup quota of 16 cores made the round-4 record 8.5x slow)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd._native import solve_batch
from oracle import likelihood as olik
from oracle.kernels import KernelSpec, RBFSpec
from oracle.solver import iterative_solve


_ORACLE = None


def _oracle_solve(rhs):
    t0 = time.perf_counter()
    xo, ito, erro, _ = iterative_solve(_ORACLE.matvec, rhs, tol=1e-4)
    return xo, ito, erro, time.perf_counter() - t0


def main_c5():
    """C5 at the reference's noise level (eps = 0.1): y and two probes solved by
    the oracle to the reference's rule (runlmc/approx/iterative.py:36-58) in
    three processes -- BEFORE the GPU is touched, so that forking is safe --,
    then all 129 systems on the device."""
    global _ORACLE
    import multiprocessing as mp
    D, Q, R, m, npr = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m, eps=0.1)
    rng = np.random.RandomState(1)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(npr)])
    spec = KernelSpec(p.D, [RBFSpec(g) for g in p.inv_lengthscales], p.coreg_vecs,
                      p.coreg_diags, p.noise)
    spec.set_input_dim(1)
    _ORACLE = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    with mp.get_context('fork').Pool(3) as pool:
        ora = pool.map(_oracle_solve, [B[0], B[1], B[2]])
    print('== c5 eps=0.1: n=%d, %d right-hand sides (y + %d probes), noise min %.3g, ||b|| = %.4g / %.4g'
          % (p.n, len(B), npr, float(np.min(p.noise)), np.linalg.norm(B[0]), np.linalg.norm(B[1])))
    print('oracle (reference rule, check every 100): iterations %s, residuals %s, %.1f s per system'
          % ([o[1] for o in ora], ['%.4g' % o[2] for o in ora], np.mean([o[3] for o in ora])))
    print('oracle seconds per iteration and system: %.4f'
          % np.mean([o[3] / max(o[1], 1) for o in ora]))
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    op = K.device_operator()
    Bd = torch.from_numpy(B).to(op.device)
    for label, gate in (('polynomial form', -1), ('transform kernels', 1 << 62)):
        op.grid.set_form_gate(gate)
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            X, it, rs, st = solve_batch(op, Bd, tol=1e-4)[:4]
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        it, rs, st = np.array(it), np.array(rs), np.array(st)
        X3 = X[:3].cpu().numpy()
        dx = [np.linalg.norm(X3[v] - ora[v][0]) / np.linalg.norm(ora[v][0]) for v in range(3)]
        print('device, %s: iterations of the three systems %s (all 129: min/mean/max %d/%.1f/%d), '
              'exit codes %s, residuals %s (all: max %.4g), %.3f s for all 129 systems = %.2f ms '
              'per iteration round; iterate rel. difference to the oracle %s'
              % (label, [int(i) for i in it[:3]], it.min(), it.mean(), it.max(),
                 sorted(set(int(s) for s in st)), ['%.4g' % r for r in rs[:3]], rs.max(),
                 min(ts), 1e3 * min(ts) / it.max(), ['%.2g' % d for d in dx]))


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    if cfg == 'c5':
        return main_c5()
    D, Q, R, m, npr = synth.CONFIGS[cfg]
    n_cpu = 3            # right-hand sides the oracle solves (one core)
    for eps in (0.1, 1.0):
        p = synth.make_problem(D, Q, R, m, eps=eps)
        fk = synth.functional_kernel(p)
        ad = (0,)
        K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        op = K.device_operator()
        rng = np.random.RandomState(1)
        B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(npr)])
        Bd = torch.from_numpy(B).to(op.device)
        print('== %s eps=%g: n=%d, %d right-hand sides (y + %d probes), noise min %.3g' % (
            cfg, eps, p.n, len(B), npr, float(np.min(p.noise))))
        dev = {}
        for ce in (100, 1):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                X, it, rs, st = solve_batch(op, Bd, tol=1e-4, check_every=ce)[:4]
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            dev[ce] = (X.cpu().numpy(), np.array(it), np.array(rs), np.array(st))
            print('device  check every %3d: iterations min/mean/max %d/%.1f/%d, residual max %.3g, '
                  'exit codes %s, %.2f ms for all %d systems' % (
                      ce, it.min(), it.mean(), it.max(), rs.max(),
                      sorted(set(int(s) for s in st)), 1e3 * min(ts), len(B)))
        spec = KernelSpec(p.D, [RBFSpec(g) for g in p.inv_lengthscales], p.coreg_vecs,
                          p.coreg_diags, p.noise)
        spec.set_input_dim(1)
        oop = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
        for ce in (100, 1):
            its, errs, secs, dx = [], [], [], []
            for v in range(n_cpu):
                t0 = time.perf_counter()
                xo, ito, erro, _ = iterative_solve(oop.matvec, B[v], tol=1e-4, check_every=ce)
                secs.append(time.perf_counter() - t0)
                its.append(ito); errs.append(erro)
                xd = dev[ce][0][v]
                dx.append(np.linalg.norm(xd - xo) / max(np.linalg.norm(xo), 1e-300))
            print('oracle  check every %3d: iterations %s (device %s), residuals %s (device %s), '
                  '%.2f s per system on one core; iterate rel. difference max %.2g' % (
                      ce, its, [int(i) for i in dev[ce][1][:n_cpu]],
                      ['%.3g' % e for e in errs], ['%.3g' % e for e in dev[ce][2][:n_cpu]],
                      float(np.mean(secs)), max(dx)))


if __name__ == '__main__':
    main()
