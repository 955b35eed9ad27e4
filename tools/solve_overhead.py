"""Fixed cost of one rl_solve_batch call (allocation, graph capture, host
round trips) against its per-round cost, on the FX2007-sized fixture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import parity_suite as ps
from cases import Case
from runlmc_amd._native import solve_batch
c = Case(sys.argv[1] if len(sys.argv) > 1 else 'fx2007')
fk, K, gk = ps.build_operator(c)
op = K.device_operator()
rng = np.random.RandomState(0)
B = torch.from_numpy(np.vstack([c.y] + [rng.randint(0, 2, c.n) * 2.0 - 1 for _ in range(15)])).to(op.device)
for maxiter in (1, 11, 101, 0):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        X, it, rs, st = solve_batch(op, B, tol=1e-4, maxiter=maxiter)[:4]
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print('maxiter %4d: %.2f ms per call (rounds %d)' % (maxiter, 1e3 * min(ts), int(np.max(it))))
