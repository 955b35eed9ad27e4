"""GPU box: time of one batched K_UU product in the polynomial form and on the
transform kernels, by batch size, at the C2 and C5 grids (where to put the batch
gate, rl_gridop_set_form_gate).   python tools/form_crossover.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp


def ms(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, batches in (('c2', (17, 32, 64, 128, 256, 512, 1024)), ('c5', (2, 4, 8, 16, 32, 64, 129))):
    D, Q, R, m, npr = synth.CONFIGS[name]
    p = synth.make_problem(D, Q, R, m)
    g = GridOp(D, p.m, Q)
    g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
    print(name, 'D', D, 'm', p.m, 'rank', g.form())
    for k in batches:
        X = torch.randn(k, D * p.m, dtype=torch.float64, device=g.device)
        Y = torch.empty_like(X)
        reps = 200 if k * D * p.m < 4e6 else 30
        g.set_form_gate(0)
        a = ms(lambda: g.mvm(X, out=Y), reps)
        g.set_form_gate(1 << 62)
        b = ms(lambda: g.mvm(X, out=Y), reps)
        g.set_form_gate(-1)
        print('  k %5d  elements %9d  polynomial %8.4f ms  transform %8.4f ms  ratio %.2f'
              % (k, k * D * p.m, a, b, b / a))
