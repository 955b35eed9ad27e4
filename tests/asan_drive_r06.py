"""ASan / UBSan pass (CPU build only; a script, not a test:
    python -m runlmc_amd.build --emu --asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_drive_r06.py)
over what round 6 added (emulator build): k_dz_mix / k_dz_resid / k_dz_norms / k_dz_axpy / k_dz_coeffs with the
host factorisation (rl_ski_factor, rl_solve_direct, rl_ski_project), k_lr_small_project / k_lr_small_expand on odd
and even grids and D above / below the segment count, k_lr_coeffs (rl_gridop_project), the preconditioned CG
(k_pcg_head / k_pcg_p / k_pcg_update, lr_all_coeffs on filter rows), the transposed weight table of k_sf_carries2,
the host helpers rl_probes_to_int8 (strided rows, several threads) and rl_slq_log_quadrature; the larger
preconditioner basis (hz_*, k_hz_*), the preconditioned log det's sampling and recorded conjugate gradients."""
import ctypes, os, sys
os.environ['RUNLMC_DEBUG'] = '1'
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np, torch
from runlmc_amd import _lib, build
lib = _lib.use_library(build.EMU_LIB.replace('.so', '_asan.so'))
from runlmc_amd._native import GridOp, SkiOp, solve_direct, solve_pcg, slq_quadratic_forms
from runlmc_amd._lib import host_ptr
from runlmc_amd.util import synth
from oracle import operators as ops
rng = np.random.RandomState(0)
# 1. direct solve, odd n per output, permuted rows
p = synth.make_problem(3, 2, 1, 331, kern='rbf')
tops = synth.tops(p)
perm = rng.permutation(p.n)
W = p.W.tocsr()[perm]; WT = W.transpose().tocsr(); WT.sort_indices()
g = GridOp(p.D, p.m, p.Q); g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
s = SkiOp(g, W, WT); s.set_noise(np.full(p.D, 0.07), p.lens)
print('factor', s.factor(), s.factor_mode)
B = torch.from_numpy(rng.randn(5, p.n))
X, it, rs, st = solve_direct(s, B, tol=1e-10)
print('direct', it, rs, st)
print('project', tuple(s.project(B).shape), tuple(g.project(s.apply_wt(B), 48).shape))
# 2. small-batch kernels: D = 1, 3, 10; odd / even m; ranks 24 and above
for D, m, sc in ((1, 401, 1.0), (3, 1000, 2.0), (10, 333, 1.0), (2, 2047, 9.0)):
    t = np.linspace(0, 1, m)
    tp = np.array([np.exp(-0.5 * (sc * t) ** 2), np.exp(-2 * np.sin(np.pi * t / 1.3) ** 2)])
    A = [rng.randn(1, D), rng.randn(2, D)]; kap = [np.abs(rng.randn(D)) + .1 for _ in range(2)]
    gg = GridOp(D, m, 2); gg.set_lmc(tp, A, kap)
    Xs = rng.randn(3, D * m)
    Bs = ops.coreg_mats(A, kap); T = [ops.BTTBOracle(x) for x in tp]
    ref = np.array([ops.grid_sum_matvec(Bs, T, v) for v in Xs])
    got = gg.matmat_host(Xs)
    print('small batch D %d m %d rank %d err %.2e' % (D, m, gg.form()[0], np.abs(got - ref).max() / np.abs(ref).max()))
# 3. preconditioned CG on Matern rows (filter form: k_sf_carries2's transposed table inside the operator)
pm = synth.make_problem(2, 2, 1, 301, kern='matern')
gm = GridOp(pm.D, pm.m, pm.Q); gm.set_lmc(synth.tops(pm), list(pm.coreg_vecs), list(pm.coreg_diags))
sm = SkiOp(gm, pm.W, pm.WT); sm.set_noise(pm.noise, pm.lens)
print('factor (matern)', sm.factor(), sm.factor_mode, gm.top_forms())
Bm = torch.from_numpy(np.vstack([pm.y] + [rng.randint(0, 2, pm.n) * 2.0 - 1 for _ in range(4)]))
Xm, itm, rsm, stm = solve_pcg(sm, Bm, tol=1e-8)
print('pcg', itm, rsm, stm)
# 3b. the larger preconditioner basis (96 functions here: m >= 768; hz_* in rl_solve.hip, k_hz_sums / k_hz_map /
#     k_hz_collect, the table arguments of k_rp_project<48> / k_rp_expand<48>): Matern rows alone, then mixed rows
os.environ['RUNLMC_PRECOND_HI_MIN'] = '0'
for kern, Qh, mh, rank, Dh in (('matern', 2, 1001, 96, 3), ('mix', 3, 1001, 96, 3), ('matern', 2, 1501, 144, 2)):
    # (144 functions = three blocks: the one-pass expansion k_hz_expand_mm and k_hz_collect's transposed layout)
    os.environ['RUNLMC_PRECOND_HI_RANK'] = str(rank)
    ph = synth.make_problem(Dh, Qh, 1, mh, kern=kern)
    gh = GridOp(ph.D, ph.m, ph.Q); gh.set_lmc(synth.tops(ph), list(ph.coreg_vecs), list(ph.coreg_diags))
    sh = SkiOp(gh, ph.W, ph.WT); sh.set_noise(ph.noise, ph.lens)
    print('factor (%s, larger basis, up to %d functions)' % (kern, rank), sh.factor(), sh.factor_mode, gh.top_forms())
    for nb in (2, 19):
        Bh = torch.from_numpy(np.vstack([ph.y] + [rng.randint(0, 2, ph.n) * 2.0 - 1 for _ in range(nb - 1)]))
        Xh, ith, rsh, sth = solve_pcg(sh, Bh, tol=1e-8)
        print('pcg', nb, ith.max(), rsh.max(), sorted(set(int(v) for v in sth)))
    # the preconditioned log det: square-root sampling (k_dz_scale, the second dense map), recorded CG
    from runlmc_amd._native import solve_pcg_lanczos
    Wh = torch.from_numpy(rng.randint(0, 2, (5, ph.n)) * 2.0 - 1)
    Rh, ldp = sh.precond_sample(Wh)
    Xl, itl, rsl, stl, lzl, sql = solve_pcg_lanczos(sh, Rh, tol=1e-8, cap=64)
    print('sample + recorded cg', ldp, itl.max(), rsl.max(), slq_quadratic_forms(lzl, itl, sql)[:2])
    gh.set_lmc(synth.tops(ph), [1.3 * a for a in ph.coreg_vecs], list(ph.coreg_diags))
    print('after an update', sh.factor(), sh.factor_mode, sh.precond_sample(Wh[:2])[1])
    del sh, gh
os.environ.pop('RUNLMC_PRECOND_HI_MIN')
os.environ.pop('RUNLMC_PRECOND_HI_RANK')
gm.set_form_gate(0)
Xg = rng.randn(4, pm.D * pm.m)
print('filter product (gate 0)', np.abs(gm.matmat_host(Xg)).max())
# 4. host helpers
rs = rng.randint(0, 2, (7, 70001)).astype(np.int64) * 2 - 1
view = rs[1::3]
out = np.zeros((view.shape[0], view.shape[1]), dtype=np.int8)
okf = ctypes.c_int()
lib.call('rl_probes_to_int8', ctypes.c_void_p(view.ctypes.data), view.shape[0], view.strides[0] // 8, view.shape[1],
         host_ptr(out), 5, ctypes.byref(okf))
print('probes', okf.value, np.array_equal(out, view.astype(np.int8)))
lz = np.zeros((3, 50, 2)); lz[:, :, 0] = 2 + rng.rand(3, 50); lz[:, :49, 1] = 0.2 * rng.rand(3, 49)
print('slq', slq_quadratic_forms(lz, np.array([50, 17, 1]), np.array([3.0, 3.0, 3.0])))
print('ASAN DRIVE DONE')
