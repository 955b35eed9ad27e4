"""GPU box: host-side timeline of ApproxLMCLikelihood._partials on the FX2007
model (where the gradient phase of a small fit goes).  python tools/partials_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np, torch
import cProfile, pstats
import fit_real_data as F
from runlmc_amd.models.interpolated_llgp import InterpolatedLLGP

name = sys.argv[1] if len(sys.argv) > 1 else 'fx2007'
xss, yss, txs, tys = F.load(name)
np.random.seed(1234)
fk, m, opt_opts, model_opts = F.kernel_for(name, len(xss))
lmc = InterpolatedLLGP(xss, yss, functional_kernel=fk, normalize=True, m=m, **model_opts)
for _ in range(3):
    lmc.parameters_changed()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    lmc.parameters_changed()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
