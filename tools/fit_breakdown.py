"""Where an end-to-end fit spends its time (GPU box): per AdaDelta step the
operator update, the probe solves (rounds), the gradient partial sums and the
host-side rest.   python tools/fit_breakdown.py [fx2007|weather|synth]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np, torch
import fit_real_data as F
from runlmc_amd.models.interpolated_llgp import InterpolatedLLGP
from runlmc_amd.models.optimization import AdaDelta
from runlmc_amd.lmc import likelihood as L
from runlmc_amd.lmc import stochastic_deriv as S

name = sys.argv[1] if len(sys.argv) > 1 else 'fx2007'
xss, yss, txs, tys = F.load(name)
np.random.seed(1234)
fk, m, opt_opts, model_opts = F.kernel_for(name, len(xss))
lmc = InterpolatedLLGP(xss, yss, functional_kernel=fk, normalize=True, m=m, **model_opts)
acc = dict(solve=0.0, parts=0.0, steps=0, rounds=0)
gen0, part0 = S.StochasticDerivService.generate, L.ApproxLMCLikelihood._partials
def gen(self, K, y, rs=None):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = gen0(self, K, y, rs)
    torch.cuda.synchronize(); acc['solve'] += time.perf_counter() - t
    acc['steps'] += 1; acc['rounds'] += int(np.max(out.iterations))
    return out
def parts(self):
    fresh = self._parts is None
    torch.cuda.synchronize(); t = time.perf_counter()
    out = part0(self)
    torch.cuda.synchronize()
    if fresh: acc['parts'] += time.perf_counter() - t
    return out
S.StochasticDerivService.generate = gen
L.ApproxLMCLikelihood._partials = parts
opt = AdaDelta(**opt_opts)
torch.cuda.synchronize(); t0 = time.perf_counter()
lmc.optimize(optimizer=opt)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
n = acc['steps']
print('%s: fit %.3f s, %d likelihood evaluations, %.1f MINRES rounds each' % (name, tot, n, acc['rounds'] / max(n, 1)))
print('  per evaluation: total %.2f ms = solves %.2f ms (%.1f us per round) + gradient partial sums %.2f ms + rest (operator update, host) %.2f ms'
      % (tot / n * 1e3, acc['solve'] / n * 1e3, acc['solve'] / max(acc['rounds'], 1) * 1e6, acc['parts'] / n * 1e3,
         (tot - acc['solve'] - acc['parts']) / n * 1e3))
