"""Round 6: two preconditioned solves of a C5 family's 129 systems, for rocprofv3's kernel statistics
(tools/r06_kernel_stats.sh pcg_<kern> tools/r06_pcg_profile.py c5 <kern>)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp, SkiOp, solve_pcg
cfg, kern = sys.argv[1], sys.argv[2]
D, Q, R, m0, N = synth.CONFIGS[cfg]
p = synth.make_problem(D, Q, R, m0, kern=kern)
g = GridOp(p.D, p.m, p.Q)
g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
s = SkiOp(g, p.W, p.WT)
s.set_noise(p.noise, p.lens)
print('factor', s.factor(), 'mode', s.factor_mode)
rng = np.random.RandomState(4321)
Bd = torch.from_numpy(np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    X, it, res, st = solve_pcg(s, Bd, tol=1e-4)
    torch.cuda.synchronize()
    print('solve %d: %.4f s, iterations %d..%d, residual max %.3g' % (rep, time.perf_counter() - t0, it.min(), it.max(), res.max()))
