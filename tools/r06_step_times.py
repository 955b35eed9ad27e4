"""Round 6: the default C5 NLL + gradient step, repeated: per-repeat wall times (spread of the host-side overlap).
    python tools/r06_step_times.py [repeats]"""
import os, sys, time
os.environ.setdefault('OMP_NUM_THREADS', '1')   # as bench.py (reference bench.py:7)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
D, Q, R, m0, N = synth.CONFIGS['c5']
p = synth.make_problem(D, Q, R, m0)
fk = synth.functional_kernel(p)
ad = (0,)
K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
np.random.seed(0)
probes = np.random.randint(0, 2, (N, p.n)) * 2 - 1
svc = StochasticDerivService(None, None, N, 1e-4)
ts = []
for r in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gks[ad].update(fk, p.grid_dists)
    lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys, svc, probes=probes)
    g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(), lik.kernel_gradients(), lik.noise_gradient())
    ll = lik.log_likelihood()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('ms per repeat (first = warm-up):', ' '.join('%.1f' % (t * 1e3) for t in ts))
print('median %.2f  min %.2f  max %.2f ms' % (np.median(ts[1:]) * 1e3, min(ts[1:]) * 1e3, max(ts[1:]) * 1e3))
