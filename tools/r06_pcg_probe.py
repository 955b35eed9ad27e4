"""Round 6: conjugate gradients preconditioned by the polynomial subspace's Woodbury inverse
(rl_solve_pcg) against MINRES on the same C5 systems.   python tools/r06_pcg_probe.py c5 mix"""
import json, os, sys, time
os.environ.setdefault('OMP_NUM_THREADS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp, SkiOp, solve_pcg, solve_batch, MINRES
cfg, kern = sys.argv[1], sys.argv[2]
D, Q, R, m0, N = synth.CONFIGS[cfg]
p = synth.make_problem(D, Q, R, m0, kern=kern)
tops = synth.tops(p)
g = GridOp(p.D, p.m, p.Q)
g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
s = SkiOp(g, p.W, p.WT)
s.set_noise(p.noise, p.lens)
sync = torch.cuda.synchronize
sync(); t0 = time.perf_counter()
av, ld, cond = s.factor()
sync()
out = dict(config=cfg, kern=kern, forms=g.top_forms()[0], available=av, mode=s.factor_mode, reason=s.factor_reason,
           factor_first_s=time.perf_counter() - t0, cond=cond)
print(json.dumps(out), flush=True)
if not av:
    sys.exit(0)
ts = []
for _ in range(2):
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    sync(); t0 = time.perf_counter(); s.factor(); sync(); ts.append(time.perf_counter() - t0)
out['factor_after_update_s'] = ts
rng = np.random.RandomState(4321)
B = np.vstack([p.y] + [rng.randint(0, 2, p.n) * 2.0 - 1 for _ in range(N)])
Bd = torch.from_numpy(B).cuda()
for tol in (1e-4, 1e-6):
    for rep in range(2):
        sync(); t0 = time.perf_counter()
        X, it, res, st = solve_pcg(s, Bd, tol=tol)
        sync(); el = time.perf_counter() - t0
    g2 = GridOp(p.D, p.m, p.Q)
    g2.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
    g2.set_form_gate(1 << 60)
    s2 = SkiOp(g2, p.W, p.WT)
    s2.set_noise(p.noise, p.lens)
    r2 = (Bd[:9] - s2.mvm(X[:9].contiguous())).norm(dim=1).cpu().numpy()
    del s2, g2
    out['pcg_tol_%g' % tol] = dict(seconds=el, iters_max=int(it.max()), iters_min=int(it.min()),
                                   resid_max=float(res.max()), resid_median=float(np.median(res)),
                                   istop=sorted(set(int(v) for v in st)),
                                   resid_transform_operator_max=float(r2.max()))
    print(json.dumps(out['pcg_tol_%g' % tol]), flush=True)
sync(); t0 = time.perf_counter()
Xk, itk, resk, stk = solve_batch(s, Bd, MINRES, tol=1e-4)[:4]
sync()
out['minres'] = dict(seconds=time.perf_counter() - t0, iters_max=int(itk.max()), resid_max=float(resk.max()),
                     resid_median=float(np.median(resk)))
out['alpha_pcg_vs_minres_rel'] = float((X[0] - Xk[0]).norm() / X[0].norm())
print(json.dumps(out))
