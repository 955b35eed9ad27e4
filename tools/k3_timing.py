"""In-kernel phase stamps of the C5 row kernel (experiment build:
python -m runlmc_amd.build --timing), workgroup (200, 1) of a 3-pair launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd import _lib
if os.environ.get('RUNLMC_LIB'):
    _lib.use_library(os.environ['RUNLMC_LIB'])
from runlmc_amd.util import synth
from runlmc_amd._native import GridOp
D, Q, R, m, npr = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c5']
nvec = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = synth.make_problem(D, Q, R, m)
g = GridOp(D, p.m, Q)
g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
X = torch.randn(nvec, D * p.m, dtype=torch.float64, device=g.device)
Y = torch.empty_like(X)
lib = _lib.get_library().cdll
if os.environ.get('K3_DBG'):
    lib.rl_debug_poke.argtypes = [ctypes.c_int, ctypes.c_longlong]
    g.mvm(X, out=Y)
    assert lib.rl_debug_poke(120, int(os.environ['K3_DBG'])) == 0
for _ in range(3):
    g.mvm(X, out=Y)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
lib.rl_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.rl_debug_timing(buf, 128) == 0
t = np.array(list(buf), dtype=np.float64) / 100.0        # 100 MHz -> microseconds
names = {60: 'start', 61: 'pass A done (thread 0)', 62: 'barrier 1 passed', 63: 'pass B done',
         64: 'barrier 2 passed', 65: 'mix done', 66: 'barrier 3 passed', 67: "pass B' done",
         68: 'barrier 4 passed', 69: "pass A' + stores issued (end)"}
for k in range(60, 70):
    print('%7.2f us  %s' % (t[k] - t[60], names[k]))
print('most workgroups of the row kernel resident at once: %d (%.2f per CU)' % (buf[101], buf[101] / 256.0))
