"""Which rank do the derivative top rows dk_q/dtheta of the benchmark's kernels verify at?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp
for cfg in ('c2', 'c5'):
    for kern in ('rbf', 'periodic'):
        D, Q, R, m0, N = synth.CONFIGS[cfg]
        p = synth.make_problem(D, Q, R, m0, kern=kern)
        fk = synth.functional_kernel(p)
        tops = synth.tops(p)
        d = fk.eval_kernel_gradients({(0,): p.grid_dists})
        dt = np.array([np.ravel(g) for q in range(Q) for g in d[q]])
        g = GridOp(D, p.m, Q)
        g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
        g2 = GridOp(D, p.m, len(dt))
        g2.set_lmc(dt, [None] * len(dt), [np.zeros(D)] * len(dt))
        g3 = GridOp(D, p.m, Q + len(dt))
        g3.set_lmc(np.vstack([tops, dt]), [None] * (Q + len(dt)), [np.zeros(D)] * (Q + len(dt)))
        print(cfg, kern, 'tops rank', g.form()[0], g.top_forms()[0], '| dtops (%d) rank' % len(dt), g2.form()[0],
              g2.top_forms()[0], '| together', g3.form()[0], g3.top_forms()[0], flush=True)
