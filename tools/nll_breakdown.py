"""Where one NLL+gradient step spends its time (GPU box): operator update,
probe solves, gradient partial sums.   python tools/nll_breakdown.py [c2|c5] [probes] [kern]
(probes: e.g. 16 = one rank's share of an 8-way split of C5's 128)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
D, Q, R, m, npr = synth.CONFIGS[cfg]
if len(sys.argv) > 2:
    npr = int(sys.argv[2])
p = synth.make_problem(D, Q, R, m, kern=sys.argv[3] if len(sys.argv) > 3 else 'rbf')
fk = synth.functional_kernel(p)
ad = (0,)
K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
svc = StochasticDerivService(None, None, npr, 1e-4)
probes = np.random.RandomState(1).randint(0, 2, (npr, p.n)) * 2 - 1


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = sync()
    gks[ad].update(fk, p.grid_dists)
    t1 = sync()
    lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys, svc,
                              probes=probes)
    t2 = sync()
    lik._partials()
    t3 = sync()
    g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(), lik.kernel_gradients(),
         lik.noise_gradient())
    t4 = sync()
    print('%s (%d probes) rep %d: update %.2f ms | solves (%d rounds) %.2f ms | partial sums %.2f ms | '
          'assembly %.2f ms | total %.2f ms' % (
              cfg, npr, rep, (t1 - t0) * 1e3, int(np.max(lik.deriv.iterations)), (t2 - t1) * 1e3,
              (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3))
