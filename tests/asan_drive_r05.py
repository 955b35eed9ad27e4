"""ASan / UBSan pass (CPU build only; a script, not a test:
    python -m runlmc_amd.build --emu --asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_drive_r05.py) over the kernels round 5 added or changed (emulator build):
k_sf_carries2, k_sf_scan1, k_sf_apply (exact-D loops), k_lr_pw_diff / k_lr_pw_scale with the
selector-coupled power iteration, k_minres2_ph + k_rp_expand<.., true> (the default solver round), k_spmv_w_staged_p + k_minres2_bv, the
plain expansion's ring of requests (rl_row_load / rl_row_store)."""
import os, sys
os.environ['RUNLMC_DEBUG'] = '1'
os.environ['RUNLMC_STAGED_WT'] = '1'; os.environ['RUNLMC_NO_FUSE_W'] = '1'; os.environ['RUNLMC_NO_FUSE_WT'] = '1'
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np, torch
from runlmc_amd import _lib, build
_lib.use_library(build.EMU_LIB.replace('.so', '_asan.so'))
from runlmc_amd._native import GridOp, solve_batch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from oracle import operators as ops
rng = np.random.RandomState(0)
# 1. filter form: Matern tops, odd grid length (the last chunk ends inside a pair), D = 3, rank-1 factors
D, m = 3, 1301
t = np.arange(m) * (1.0 / m)
mat = lambda g: (1 + np.sqrt(3) * g * t) * np.exp(-np.sqrt(3) * g * t)
tops = np.array([mat(2.0), mat(7.0)])
A = [rng.randn(1, D), rng.randn(1, D)]; k = [np.abs(rng.randn(D)) + .1 for _ in range(2)]
g = GridOp(D, m, 2); g.set_lmc(tops, A, k); g.set_form_gate(0)
print('filter forms', g.top_forms())
X = rng.randn(5, D * m)
Bs = ops.coreg_mats(A, k); T = [ops.BTTBOracle(tp) for tp in tops]
ref = np.array([ops.grid_sum_matvec(Bs, T, v) for v in X])
got = g.matmat_host(X)
print('filter err', np.abs(got - ref).max() / np.abs(ref).max())
# 2. polynomial verification with the grouped power iteration: Q = 5 tops > D = 2 outputs (three groups)
D2, m2 = 2, 700
t2 = np.linspace(0, 1, m2)
tops2 = np.array([np.exp(-0.5 * (a * t2) ** 2) for a in (1.0, 1.5, 2.0, 2.5, 3.0)])
g2 = GridOp(D2, m2, 5)
g2.set_lmc(tops2, [rng.randn(1, D2) for _ in range(5)], [np.abs(rng.randn(D2)) + .1 for _ in range(5)])
g2.set_form_gate(0)
print('poly forms', g2.top_forms(), g2.form(), [g2.form_stats(q)[2:] for q in range(5)])
# 3. MINRES with P inside the expansion
p = synth.make_problem(3, 2, 1, 700, eps=1.0, kern='rbf')
fk = synth.functional_kernel(p); ad = (0,)
K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
op = K.device_operator(); op.grid.set_form_gate(0)
V = rng.randn(19, p.n)
out = solve_batch(op, torch.from_numpy(V).to(op.device), tol=1e-3, maxiter=60, lanczos_cap=8)
print('pfuse iterations', np.asarray(out[1]), 'istop', np.asarray(out[3]))
# 4. MINRES with P inside the staged W product (k_spmv_w_staged_p, k_minres2_bv): Matern tops, ragged
#    outputs, 11 systems (a full group of eight and three), a solve that ends (frozen systems)
p = synth.make_problem(3, 2, 1, 500, eps=1.0, kern='matern')
fk = synth.functional_kernel(p)
K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
op = K.device_operator(); op.grid.set_form_gate(0)
V = rng.randn(11, p.n)
out = solve_batch(op, torch.from_numpy(V).to(op.device), tol=1e-3, maxiter=200, lanczos_cap=8)
print('P-in-W iterations', np.asarray(out[1]), 'istop', np.asarray(out[3]))
# 5. the plain expansion with the noise term (rl_ski_mvm: ring of three requests, buffer accesses) at a
#    row count that is no multiple of 256 and a batch that is no multiple of 3
p = synth.make_problem(3, 2, 1, 700, eps=1.0, kern='rbf')
fk = synth.functional_kernel(p)
K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
op = K.device_operator(); op.grid.set_form_gate(0)
Y = op.matmat_host(rng.randn(7, p.n))
print('row-polynomial product', np.abs(Y).max())
print('ASAN DRIVE DONE')
