"""The single-tile product k1_product<D> (short grids: the FX2007 / weather shapes) for
profilers:  python tools/k1_product_run.py [D] [m] [batch] [calls]
Default: the FX2007 shape (D = 13, m = 238) on 256 vectors.  Prints us per product."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd._native import GridOp      # noqa: E402
from runlmc_amd.util import synth          # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 13
m = int(sys.argv[2]) if len(sys.argv) > 2 else 238
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rng = np.random.RandomState(0)
x = np.linspace(0, 1, m)
# (a short length scale: the polynomial form rejects it, and the batch stays below the
# structured forms' gate anyway)
tops = np.array([np.exp(-0.5 * 4000.0 * x ** 2)])
g = GridOp(D, m, 1)
g.set_lmc(tops, [rng.randn(2, D)], [np.abs(rng.randn(D)) + 0.1])
g.set_form_gate(1 << 62)
X = torch.randn(batch, D * m, dtype=torch.float64, device=g.device)
Y = torch.empty_like(X)
for _ in range(3):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / calls * 1e6
alg = synth.algorithmic_bytes_grid_mvm(D, 1, m, g.L, batch)
print('k1_product shape D=%d m=%d L=%d, %d vectors: %.1f us per product, %.1f %% of the '
      'algorithmic HBM roofline' % (D, m, g.L, batch, us, alg / (us * 1e-6) / 8e12 * 100))
