"""GPU box: MINRES solves of a synthetic LMC problem three ways -- B inside the row-polynomial
projection (default), B as its own kernel (RUNLMC_NO_RP_FUSE=1), interpolation products
(RUNLMC_NO_RP=1): iterates, iteration counts, exit codes.   python tools/rp_fuse_check.py [nrhs] [maxiter]"""
import sys, os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, 'tests'))
os.environ['RUNLMC_STAGED_WT']='1'; os.environ['RUNLMC_NO_FUSE_W']='1'; os.environ['RUNLMC_NO_FUSE_WT']='1'
os.environ['RUNLMC_TRACE']='1'; os.environ['RUNLMC_DEBUG']='1'   # (the switches are debug hooks)
from runlmc_amd import _lib

import numpy as np, torch
from runlmc_amd.util import synth
from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
from runlmc_amd._native import solve_batch
D,Q,m_data=3,2,2600
p = synth.make_problem(D, Q, 1, m_data, eps=1.0, kern='rbf')
fk = synth.functional_kernel(p); ad=(0,)
rng=np.random.RandomState(1)
nv=int(sys.argv[1]) if len(sys.argv)>1 else 3
mi=int(sys.argv[2]) if len(sys.argv)>2 else 6
V=rng.randn(nv,p.n)
res={}
for mode in ('fused','unfused','norp'):
    for k in ('RUNLMC_NO_RP','RUNLMC_NO_RP_FUSE'): os.environ.pop(k,None)
    if mode=='unfused': os.environ['RUNLMC_NO_RP_FUSE']='1'
    if mode=='norp': os.environ['RUNLMC_NO_RP']='1'
    K,_=gen_grid_kernel(fk,{ad:p.grid_dists},{ad:(p.W,p.WT)},p.lens)
    op=K.device_operator(); op.grid.set_form_gate(0)
    X,it,rs,st=solve_batch(op, torch.from_numpy(V).to(op.device), tol=float(os.environ.get("DBG_TOL","1e-6")), maxiter=mi)[:4]
    res[mode]=(X.cpu().numpy(),np.asarray(it),np.asarray(rs),np.asarray(st))
    print(mode, res[mode][1][:8], res[mode][3][:8], res[mode][2][:4])
a,b,c=res['fused'][0],res['unfused'][0],res['norp'][0]
print('fused vs unfused', np.abs(a-b).max()/np.abs(b).max(), 'unfused vs norp', np.abs(b-c).max()/np.abs(c).max())
print('iters equal', np.array_equal(res['fused'][1],res['unfused'][1]), np.array_equal(res['fused'][3],res['unfused'][3]))
