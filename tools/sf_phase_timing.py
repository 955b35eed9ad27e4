"""Phase stamps of one mid-launch workgroup of k_sf_apply (timing build:
python -m runlmc_amd.build --timing): python tools/sf_phase_timing.py [kern]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd import _lib
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp
kern = sys.argv[1] if len(sys.argv) > 1 else 'matern'
D, Q, R, m_data, n_probes = synth.CONFIGS['c5']
p = synth.make_problem(D, Q, R, m_data, kern=kern)
g = GridOp(D, p.m, Q)
g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
g.set_form_gate(0)
X = torch.randn(n_probes + 1, D * p.m, dtype=torch.float64, device='cuda')
Y = torch.empty_like(X)
for _ in range(3):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
lib = _lib.get_library().cdll
buf = (ctypes.c_longlong * 256)()
lib.rl_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.rl_debug_timing(buf, 256) == 0
t = np.array(list(buf), dtype=np.float64) / 100.0        # 100 MHz -> microseconds
names = [(99, 'recurrences start'), (103, 'rows of x done'), (104, 'next tile requested'),
         (110, 'mixed rows done'), (111, 'after the barrier'), (112, 'y assembled and stored'),
         (100, 'next tile: staging starts'), (101, 'rows + incoming states in LDS'),
         (102, 'mixed rows formed, after the barrier')]
for wave in range(4):
    base = t[99 + 30 * wave]
    print('wave %d of workgroup 100, its 21st tile:' % wave)
    for k, label in names:
        if t[k + 30 * wave] > 0:
            print('  %8.2f us  %s' % (t[k + 30 * wave] - base, label))
print('resident workgroups at most:', int(buf[121]))
if buf[212]:
    print('workgroup lifetimes over %d workgroups: shortest %.1f us, mean %.1f us, longest %.1f us'
          % (buf[212], ((1 << 40) - buf[213]) / 100.0, buf[211] / buf[212] / 100.0, buf[210] / 100.0))
