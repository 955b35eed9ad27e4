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
names = {100: 'start', 101: 'x tile + tables in LDS', 102: 'mixed rows + block tops done',
         103: 'pass 0', 104: 'pass 1', 105: 'pass 2', 106: 'pass 3', 107: 'pass 4', 108: 'pass 5',
         110: 'passes done (thread 0)', 111: 'after barrier', 112: 'stored'}
for k in sorted(names):
    if t[k] > 0:
        print('%8.2f us  %s' % (t[k] - t[100], names[k]))
print('resident workgroups at most:', int(buf[121]))
