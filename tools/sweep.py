"""Grid-product throughput over the (D, Q, m) sweep SURVEY section 8d lists
(m in 1e3, 1e4, 1e5; D in 2, 4, 10; Q in 1, 3, 5), at the probe batch (17
vectors) and at a saturating batch, with the FORM each product ran in (poly =
polynomial-subspace form, filter = recursive filter, fft = transform kernels) and
the saturating batch again on the transform kernels.
GPU box:  python tools/sweep.py [rbf|matern] > table.txt"""
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


kern = sys.argv[1] if len(sys.argv) > 1 else 'rbf'


def form_of(g, k, D, m):
    forms, structured = g.top_forms()
    big = k * D * m >= g.form()[1]
    if not (big and structured):
        return 'fft'
    return 'poly' if all(f == 1 for f in forms) else ('filter' if all(f == 2 for f in forms)
                                                      else 'mixed')


print('kernel family:', kern)
print('%3s %2s %7s %7s | %11s %7s %6s | %6s %11s %7s %6s | %11s %7s' % (
    'D', 'Q', 'm', 'L', 'MVM/s k=17', 'roofl', 'form', 'k_sat', 'MVM/s', 'roofl', 'form',
    'fft MVM/s', 'roofl'))
rng = np.random.RandomState(0)
for m in (1000, 10000, 100000):
    for D in (2, 4, 10):
        for Q in (1, 3, 5):
            xg = np.arange(m) / m * 3.0
            if kern == 'matern':
                tops = np.array([(1 + np.sqrt(3) * gq * xg) * np.exp(-np.sqrt(3) * gq * xg)
                                 for gq in np.logspace(0, 1, Q)])
            else:
                tops = np.array([np.exp(-0.5 * xg ** 2 * gq) for gq in np.logspace(0, 1, Q)])
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
                row.append((k / s, ab / s / 1e9 / HBM, form_of(g, k, D, m)))
                if k == ksat:
                    g.set_form_gate(1 << 62)
                    sf = timed(g, X, Y, 20 if k * D * m < 5e7 else 5)
                    g.set_form_gate(-1)
                    row.append((k / sf, ab / sf / 1e9 / HBM, 'fft'))
                del X, Y
            print('%3d %2d %7d %7d | %11.0f %6.2f%% %6s | %6d %11.0f %6.2f%% %6s | %11.0f %6.2f%%' % (
                D, Q, m, g.L, row[0][0], 100 * row[0][1], row[0][2], ksat, row[1][0],
                100 * row[1][1], row[1][2], row[2][0], 100 * row[2][1]), flush=True)
            del g
