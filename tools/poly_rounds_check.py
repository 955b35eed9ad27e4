"""GPU box: polynomial rounds against the transform rounds at larger D / Q than the
parity suite uses (iterates after six iterations and at the exit).  python tools/poly_rounds_check.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd._native import solve_batch
for D, Q, m, k in ((12, 4, 2100, 5), (16, 2, 2300, 3), (7, 5, 3000, 9)):
    p = synth.make_problem(D, Q, 1, m, eps=1.0)
    p.noise = p.noise + 1.0
    fk = synth.functional_kernel(p)
    ad = (0,)
    rng = np.random.RandomState(5)
    B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(k - 1)])
    res = {}
    for mode in ('poly', 'fft'):
        if mode == 'fft':
            os.environ['RUNLMC_NO_POLY_ROUND'] = '1'
        else:
            os.environ.pop('RUNLMC_NO_POLY_ROUND', None)
        K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
        op = K.device_operator()
        out6 = solve_batch(op, torch.from_numpy(B).to(op.device), tol=1e-4, maxiter=6)
        out = solve_batch(op, torch.from_numpy(B).to(op.device), tol=1e-4)
        res[mode] = (out6[0].cpu().numpy(), out[0].cpu().numpy(), np.array(out[1]), op.grid.form())
    a6, a, ia, fa = res['poly']; b6, b, ib, fb = res['fft']
    rel6 = np.abs(a6 - b6).max() / np.abs(b6).max()
    rel = np.abs(a - b).max() / np.abs(b).max()
    print('D %d Q %d m %d k %d  form %s  iterate@6 rel %.2e  converged rel %.2e  iters %s vs %s' % (D, Q, m, k, fa, rel6, rel, ia, ib))
    # (at the exit the two paths may stop a few iterations apart: SciPy's test1 at its roundoff floor)
    assert rel6 < 1e-8 and rel < 5e-4
print('ok')
