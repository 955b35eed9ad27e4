"""Timing probe of the grid product at the C2 grid size, one batch size (GPU box).

    python tools/v4_probe.py [batch]

Environment: RUNLMC_V4_MIN=<batch> turns the on-chip product on, RUNLMC_NO_V4=1
removes it, RUNLMC_TILE_C / RUNLMC_TILE_R / RUNLMC_THR_C / RUNLMC_THR_R override
the tile choice of the three-kernel path (tools/tile_sweep.sh).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd._native import GridOp
D, Q, m = 4, 3, 5004
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.RandomState(0)
tops = np.array([np.exp(-(0.02 + 0.1 * q) * np.arange(m)) for q in range(Q)])
g = GridOp(D, m, Q)
g.set_lmc(tops, [rng.randn(1, D) for _ in range(Q)], [np.abs(rng.randn(D)) + .1 for _ in range(Q)])
X = torch.randn(b, D * m, dtype=torch.float64, device=g.device)
Y = torch.empty_like(X)
for _ in range(3):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 20
for _ in range(n):
    g.mvm(X, out=Y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print('batch=%d onchip=%s  %.1f us/product  %.2f M MVM/s  (%.1f us of one CU per vector)' % (
    b, g.onchip, ms * 1e3, b / ms / 1e3, ms * 1e3 / max(1, b / 256)))
