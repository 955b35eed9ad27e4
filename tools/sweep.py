"""Grid-product throughput over the (D, Q, m) sweep SURVEY section 8d lists
(m in 1e3, 1e4, 1e5; D in 2, 4, 10; Q in 1, 3, 5), at the probe batch (17
vectors) and at a saturating batch.  GPU box:  python tools/sweep.py > table.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd._native import GridOp
from runlmc_amd.util import synth

HBM = 8000.0


def timed(g, X, Y, reps):
    for _ in range(3):
        g.mvm(X, out=Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.mvm(X, out=Y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print('%3s %2s %7s %7s | %11s %7s | %6s %11s %7s' % (
    'D', 'Q', 'm', 'L', 'MVM/s k=17', 'roofl', 'k_sat', 'MVM/s', 'roofl'))
rng = np.random.RandomState(0)
for m in (1000, 10000, 100000):
    for D in (2, 4, 10):
        for Q in (1, 3, 5):
            tops = np.array([np.exp(-0.5 * (np.arange(m) / m * 3.0) ** 2 * gq)
                             for gq in np.logspace(0, 1, Q)])
            g = GridOp(D, m, Q)
            g.set_lmc(tops, [rng.randn(1, D) for _ in range(Q)],
                      [np.abs(rng.randn(D)) + 0.1 for _ in range(Q)])
            row = []
            ksat = int(max(32, min(4096, 2 ** int(np.log2(6e8 / (D * m * 8))))))
            for k in (17, ksat):
                X = torch.randn(k, D * m, dtype=torch.float64, device=g.device)
                Y = torch.empty_like(X)
                s = timed(g, X, Y, 20 if k * D * m < 5e7 else 5)
                ab = synth.algorithmic_bytes_grid_mvm(D, Q, m, g.L, k)
                row.append((k / s, ab / s / 1e9 / HBM))
                del X, Y
            print('%3d %2d %7d %7d | %11.0f %6.2f%% | %6d %11.0f %6.2f%%' % (
                D, Q, m, g.L, row[0][0], 100 * row[0][1], ksat, row[1][0], 100 * row[1][1]),
                flush=True)
            del g
