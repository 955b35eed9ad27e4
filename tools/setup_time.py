import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp
for name in ('c2', 'c5'):
    D, Q, R, m, npr = synth.CONFIGS[name]
    p = synth.make_problem(D, Q, R, m)
    g = GridOp(D, p.m, Q)
    tops = synth.tops(p)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        r = g.form()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(name, 'set %.3f ms  verification %.3f ms  rank %s' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, r))
