"""In-kernel phase stamps of the last solver round (experiment build:
python -m runlmc_amd.build --timing).  Prints microseconds relative to the
start of k2_cols_fwd, workgroup (0,0,0) only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd import _lib
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd._native import solve_batch
D, Q, R, m, npr = synth.CONFIGS['c2']
p = synth.make_problem(D, Q, R, m)
fk = synth.functional_kernel(p)
K, _ = gen_grid_kernel(fk, {(0,): p.grid_dists}, {(0,): (p.W, p.WT)}, p.lens)
op = K.device_operator()
rng = np.random.RandomState(3)
B = torch.from_numpy(rng.randint(0, 2, (npr + 1, p.n)) * 2.0 - 1).to(op.device)
solve_batch(op, B, tol=1e-4, maxiter=41)
lib = _lib.get_library().cdll
buf = (ctypes.c_longlong * 64)()
lib.rl_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.rl_debug_timing(buf, 64) == 0
t = np.array(list(buf), dtype=np.float64) / 100.0        # 100 MHz -> microseconds
names = {10: 'cols_fwd start', 11: 'cols_fwd first pass done (gather + radix)', 12: 'cols_fwd middle done',
         13: 'cols_fwd end', 20: 'rows_mix start', 22: 'rows_mix first pass done (loads + radix)', 23: 'rows_mix forward done', 24: 'rows_mix mix done', 25: 'rows_mix adjoint LDS passes done', 21: 'rows_mix end', 32: 'cols_inv first pass done', 33: 'cols_inv middle done', 30: 'cols_inv start',
         31: 'cols_inv end', 0: 'P start', 1: 'P operands requested', 2: 'P partial sums reduced',
         3: 'P scalars done', 4: 'P vector part done', 5: 'P dots reduced', 6: 'P end',
         40: 'B start', 41: 'B end', 7: 'P poly: prologue start', 8: 'P poly: partial sums in LDS',
         9: 'P poly: coefficients mixed', 42: 'B y done', 43: 'B poly: rows accumulated', 44: 'B poly: block sums done'}
poly = t[7] > 0          # (polynomial rounds: no grid kernels)
t0 = t[0] if poly else t[10]
order = (0, 1, 7, 8, 9, 2, 3, 4, 5, 6, 40, 42, 43, 44, 41) if poly else \
    (10, 11, 12, 13, 20, 22, 23, 24, 25, 21, 30, 32, 33, 31, 0, 1, 2, 3, 4, 5, 6, 40, 41)
for k in order:
    print('%7.2f us  %s' % (t[k] - t0, names[k]))
