"""One kernel family's grid product with the library's DEFAULT gate (the form a caller gets for this
batch), a few calls (for profilers):  python tools/one_family_default.py c2 rbf 17 50 [fft]
(fft: the same product on the transform kernels)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd.util import synth          # noqa: E402
from runlmc_amd._native import GridOp      # noqa: E402
cfg, kern = sys.argv[1], sys.argv[2]
D, Q, R, m_data, n_probes = synth.CONFIGS[cfg]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else n_probes + 1
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 5
p = synth.make_problem(D, Q, R, m_data, kern=kern)
g = GridOp(D, p.m, Q)
g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
if len(sys.argv) > 5 and sys.argv[5] == 'fft':
    g.set_form_gate(1 << 62)
X = torch.randn(batch, D * p.m, dtype=torch.float64, device='cuda')
Y = torch.empty_like(X)
for _ in range(calls):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
print(g.top_forms(), g.form())
