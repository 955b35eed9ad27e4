"""Scratch: first timings of the grid MVM on the GPU (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from runlmc_amd._native import GridOp
def run(D,Q,m,nvec,reps=20):
    rng=np.random.RandomState(0)
    tops=np.array([np.exp(-0.5*(q+1)*np.linspace(0,1,m)**2*100) for q in range(Q)])
    A=[rng.randn(1,D) for _ in range(Q)]; kap=[np.abs(rng.randn(D)) for _ in range(Q)]
    g=GridOp(D,m,Q); g.set_lmc(tops,A,kap)
    X=torch.randn(nvec,D*m,dtype=torch.float64,device=g.device); Y=torch.empty_like(X)
    for _ in range(3): g.mvm(X,out=Y)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.mvm(X,out=Y)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    alg=8*(2*D*m*nvec+Q*(g.L//2+1))
    print(f'D={D} Q={Q} m={m} L={g.L} N1={g.N1} N2={g.N2} C={g.colsA} R={g.rowsB} nvec={nvec}: {ms*1e3:.1f} us/batch  {nvec/ms*1e3:.0f} MVM/s  alg {alg/ms/1e6:.1f} GB/s ({alg/ms/1e6/8000*100:.2f}% of 8TB/s)')
for nvec in (2,17,64,256,1024):
    run(4,3,5004,nvec)
for nvec in (2,16,129):
    run(10,5,100004,nvec,reps=5)
run(13,1,238,16); run(4,6,1004,16); run(2,2,104,16)
