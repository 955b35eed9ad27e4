"""Round 6: a small batch of the C2 operator -- the one-launch polynomial product (k_lr_small)
against the transform kernels of the same handle.   python tools/r06_small_batch.py [nvec ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp

cfg = 'c2'
D, Q, R, m0, N = synth.CONFIGS[cfg]
kern = os.environ.get('KERN', 'rbf')
p = synth.make_problem(D, Q, R, m0, kern=kern)
g = GridOp(p.D, p.m, p.Q)
g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
for nvec in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 17, 32, 48]:
    X = torch.randn(nvec, D * p.m, dtype=torch.float64, device='cuda')
    Y = torch.empty_like(X)
    res = {}
    for mode, gate in (('default', -1), ('transform', 1 << 60), ('three-launch polynomial', 0)):
        g.set_form_gate(gate)
        for _ in range(20):
            g.mvm(X, out=Y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(500):
            g.mvm(X, out=Y)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 500 * 1e6, e0.elapsed_time(e1) / 500 * 1e3, Y.clone())
    g.set_form_gate(-1)
    d = float((res['default'][2] - res['transform'][2]).abs().max() / res['transform'][2].abs().max())
    alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, nvec)
    print('%s %s nvec %3d: default %.2f us (events %.2f) = %.1f %% of 8 TB/s | transform kernels %.2f us | '
          'three-launch polynomial %.2f us | default vs transform %.1e | rank %d'
          % (cfg, kern, nvec, res['default'][0], res['default'][1], alg / res['default'][0] / 8e6 * 100,
             res['transform'][0], res['three-launch polynomial'][0], d, g.form()[0]), flush=True)
