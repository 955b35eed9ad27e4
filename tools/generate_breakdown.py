"""Where StochasticDerivService.generate spends a C5 step outside the solver's rounds
(GPU box):  python tools/generate_breakdown.py [rbf|periodic|matern|mix]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd.approx.iterative import Iterative

D, Q, R, m, npr = synth.CONFIGS['c5']
p = synth.make_problem(D, Q, R, m, kern=sys.argv[1] if len(sys.argv) > 1 else 'rbf')
fk = synth.functional_kernel(p)
ad = (0,)
K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
rs = (np.random.RandomState(1).randint(0, 2, (npr, p.n)) * 2 - 1).astype(np.int8)
dev = K.device


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = sync()
    narrow = torch.from_numpy(rs).to(torch.int8).to(dev)
    t1 = sync()
    ok = bool((narrow.abs() == 1).all())
    t2 = sync()
    B = torch.zeros((npr + 1, p.n), dtype=torch.float64, device=dev)
    B[npr] = torch.from_numpy(np.ascontiguousarray(p.y, dtype=np.float64)).to(dev)
    B[:npr] = narrow
    t3 = sync()
    X, it, res, istop, lz = Iterative.solve_device(K, B, minres=True, tol=1e-4, lanczos_cap=256)
    t4 = sync()
    idx = torch.tensor([npr] + list(range(npr)), device=dev)
    X2, B2 = X[idx], B[idx]
    t5 = sync()
    print('rep %d: probes to the device %.1f ms | +-1 check %.1f | assemble %.1f | solve_device %.1f '
          '(%d rounds -> %.3f ms per round) | reorder %.1f' % (
              rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3),
              int(np.max(it)), 1e3 * (t4 - t3) / int(np.max(it)), 1e3 * (t5 - t4)))
