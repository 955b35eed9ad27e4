"""Per-round cost of the batched MINRES solve on a synthetic config (GPU box).
   python tools/solve_rounds.py [c2|c5] [nrhs] [maxiter]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd._native import solve_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 43
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 41
D, Q, R, m, npr = synth.CONFIGS[cfg]
p = synth.make_problem(D, Q, R, m, kern=sys.argv[4] if len(sys.argv) > 4 else "rbf")
fk = synth.functional_kernel(p)
ad = (0,)
K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
op = K.device_operator()
rng = np.random.RandomState(0)
B = torch.from_numpy(rng.randint(0, 2, (nrhs, p.n)) * 2.0 - 1).to(op.device)
for mi in (1, maxiter):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        X, it, rs, st = solve_batch(op, B, tol=1e-4, maxiter=mi)[:4]
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print('%s nrhs %d maxiter %3d: %.2f ms per call (rounds %d)' % (
        cfg, nrhs, mi, 1e3 * min(ts), int(np.max(it))), flush=True)
    if mi == 1:
        t1 = min(ts)
print('per round: %.3f ms' % (1e3 * (min(ts) - t1) / (maxiter - 1)))
