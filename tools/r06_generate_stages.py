"""Where does StochasticDerivService.generate spend its time at C5 (direct solves)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService, _host_cores
from runlmc_amd._native import solve_direct

D, Q, R, m, npr = synth.CONFIGS['c5']
p = synth.make_problem(D, Q, R, m)
fk = synth.functional_kernel(p)
ad = (0,)
K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
probes = np.random.RandomState(1).randint(0, 2, (npr, p.n)) * 2 - 1
dev = K.device


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


print('torch threads', torch.get_num_threads(), 'host cores', _host_cores(), 'OMP', os.environ.get('OMP_NUM_THREADS'))
for rep in range(3):
    t = [sync()]
    wide = torch.from_numpy(probes); t.append(sync())
    torch.set_num_threads(min(_host_cores(), 32))
    lo, hi = torch.aminmax(wide); t.append(sync())
    n8 = wide.to(torch.int8); t.append(sync())
    torch.set_num_threads(1)
    nd = n8.to(dev); t.append(sync())
    ok = bool((nd != 0).all()); t.append(sync())
    B = torch.zeros((npr + 1, p.n), dtype=torch.float64, device=dev); t.append(sync())
    B[:npr] = nd; t.append(sync())
    B[npr] = torch.from_numpy(p.y).to(dev); t.append(sync())
    gks[ad].update(fk, p.grid_dists); t.append(sync())
    M = K.preconditioner; t.append(sync())
    X = M.solve(B, tol=1e-4); t.append(sync())
    ld = M.logdet(); t.append(sync())
    names = ['from_numpy', 'aminmax', 'to int8', 'H2D', 'check', 'zeros', 'widen', 'y', 'update', 'preconditioner (verify + factor)', 'solve', 'logdet']
    print('rep', rep, ' | '.join('%s %.2f' % (n, (b - a) * 1e3) for n, a, b in zip(names, t[:-1], t[1:])), 'ms')
svc = StochasticDerivService(None, None, npr, 1e-4)
for rep in range(3):
    t0 = sync()
    d = svc.generate(K, p.y, rs=probes)
    print('generate %.2f ms' % ((sync() - t0) * 1e3))
