"""Grid product on short grids (the reference's real-data sizes): single-tile
kernel vs the three-kernel path, microseconds per batched product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd._native import GridOp
rng = np.random.RandomState(0)


def timed(g, X, Y, reps=50):
    for _ in range(5):
        g.mvm(X, out=Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.mvm(X, out=Y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for D, Q, m in ((2, 1, 104), (13, 1, 238), (4, 6, 504), (4, 6, 1004), (8, 3, 600), (16, 2, 300)):
    tops = np.array([np.exp(-0.5 * (np.arange(m) / m * 3.0) ** 2 * gq) for gq in np.logspace(0, 1, Q)])
    A = [rng.randn(1, D) for _ in range(Q)]
    kap = [np.abs(rng.randn(D)) + 0.1 for _ in range(Q)]
    out = []
    for mode in ('three-kernel', 'single-tile'):
        if mode == 'three-kernel':
            os.environ['RUNLMC_NO_V1P'] = '1'
        else:
            os.environ.pop('RUNLMC_NO_V1P', None)
        g = GridOp(D, m, Q)
        g.set_lmc(tops, A, kap)
        row = []
        for k in (1, 16, 64, 256, 2048):
            X = torch.randn(k, D * m, dtype=torch.float64, device=g.device)
            Y = torch.empty_like(X)
            row.append(timed(g, X, Y, 50 if k < 2048 else 10))
        out.append((mode, row))
    print('D=%2d Q=%d m=%5d L=%5d | ' % (D, Q, m, g.L) +
          ' | '.join('%s: %s' % (mo, ' '.join('%7.1f' % t for t in r)) for mo, r in out), flush=True)
