"""Round 6: the direct solve (csrc/rl_direct.h) on the GPU -- factorisation time, solve time,
residuals through the handle's product AND through an independent transform-kernel operator,
against the Krylov solve of the same systems.   python tools/r06_direct_probe.py c5 rbf"""
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from runlmc_amd.util import synth
from runlmc_amd._native import GridOp, SkiOp, solve_direct, solve_batch, MINRES

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
kern = sys.argv[2] if len(sys.argv) > 2 else 'rbf'
D, Q, R, m0, N = synth.CONFIGS[cfg]
p = synth.make_problem(D, Q, R, m0, kern=kern)
tops = synth.tops(p)
out = dict(config=cfg, kern=kern, n=p.n, m=p.m, D=D, Q=Q)


def sync():
    torch.cuda.synchronize()


g = GridOp(p.D, p.m, p.Q)
g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
s = SkiOp(g, p.W, p.WT)
s.set_noise(p.noise, p.lens)
sync()
t0 = time.perf_counter()
av, logdet, cond = s.factor()
sync()
out['factor_first_s'] = time.perf_counter() - t0
out.update(available=av, logdet=logdet, cond=cond, rank=g.form()[0], reason=s.factor_reason)
print(json.dumps(out), flush=True)
if not av:
    sys.exit(0)
ts = []
for _ in range(3):
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    sync()
    t0 = time.perf_counter()
    s.factor()
    sync()
    ts.append(time.perf_counter() - t0)
out['factor_after_update_s'] = ts
rng = np.random.RandomState(4321)
B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])
Bd = torch.from_numpy(B).cuda()
for tol in (1e-4, 1e-6):
    rec = {}
    for rep in range(3):
        sync()
        t0 = time.perf_counter()
        X, it, res, istop = solve_direct(s, Bd, tol=tol)
        sync()
        rec.setdefault('seconds', []).append(time.perf_counter() - t0)
    rec.update(iters_max=int(it.max()), iters_min=int(it.min()), resid_max=float(res.max()),
               resid_median=float(np.median(res)), istop=sorted(set(int(v) for v in istop)))
    # the same residuals through an operator that runs on the transform kernels only
    g2 = GridOp(p.D, p.m, p.Q)
    g2.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    g2.set_form_gate(1 << 60)
    s2 = SkiOp(g2, p.W, p.WT)
    s2.set_noise(p.noise, p.lens)
    k = min(9, len(B))
    r2 = (Bd[:k] - s2.mvm(X[:k].contiguous())).norm(dim=1).cpu().numpy()
    rec['resid_transform_operator_max'] = float(r2.max())
    rec['resid_transform_operator'] = [float(v) for v in r2[:4]]
    del s2, g2
    out['direct_tol_%g' % tol] = rec
    print(json.dumps(rec), flush=True)
Xd = X
# the Krylov solve of the same systems (SciPy's exits, as the headline step until round 5)
sync()
t0 = time.perf_counter()
Xk, itk, resk, istk = solve_batch(s, Bd, MINRES, tol=1e-4)[:4]
sync()
out['minres'] = dict(seconds=time.perf_counter() - t0, iters_max=int(itk.max()),
                     resid_max=float(resk.max()), resid_median=float(np.median(resk)))
out['alpha_direct_vs_minres_rel'] = float((Xd[0] - Xk[0]).norm() / Xd[0].norm())
print(json.dumps(out))
