"""Scratch: time W^T x and W g at the C5 size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp, SkiOp
D, Q, R, m, N = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c5']
p = synth.make_problem(D, Q, R, m)
g = GridOp(D, p.m, Q); g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
s = SkiOp(g, p.W, p.WT); s.set_noise(p.noise, p.lens)
nvec = N + 1
X = torch.randn(nvec, p.n, dtype=torch.float64, device=g.device)
for name, fn in (('apply_wt', lambda: s.apply_wt(X)), ('ski_mvm', lambda: s.mvm(X))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, 'nvec', nvec, '%.3f ms' % (e0.elapsed_time(e1) / 5))
