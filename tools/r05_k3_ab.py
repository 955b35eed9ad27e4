"""GPU box: the row kernel of the transform path, k3_rows_mix<D <= 5, ..>, at 128 registers (four waves
per SIMD, 14-16 registers spilled) against a build at 168 (three waves, no spill: -DRL_K3_WPE_SMALL=3,
runlmc_amd/csrc/librunlmc_hip_k3w3.so), same box, alternating: C2 (D=4, Q=3, m=5000) products on the
transform kernels at 17 / 256 / 1024 vectors.  The second library is an experiment build, not kept in the tree:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRL_K3_WPE_SMALL=3 \
          -o runlmc_amd/csrc/librunlmc_hip_k3w3.so runlmc_amd/csrc/runlmc_hip.hip
    python tools/r05_k3_ab.py          (result: profiles/r05/k3_registers_ab.txt -- no difference)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd import _lib                # noqa: E402
from runlmc_amd.util import synth          # noqa: E402

LIBS = {'128 registers (default)': _lib.HIP_LIB,
        '168 registers (k3w3)': os.path.join(ROOT, 'runlmc_amd', 'csrc', 'librunlmc_hip_k3w3.so')}


def timeit(fn, steps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


D, Q, R, m_data, _ = synth.CONFIGS['c2']
p = synth.make_problem(D, Q, R, m_data)
res = {}
for rep in range(3):
    for name, path in LIBS.items():
        _lib.use_library(path)
        from runlmc_amd._native import GridOp
        g = GridOp(D, p.m, Q)
        g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
        g.set_form_gate(1 << 62)
        for batch in (17, 256, 1024):
            X = torch.randn(batch, D * p.m, dtype=torch.float64, device='cuda')
            Y = torch.empty_like(X)
            us = timeit(lambda: g.mvm(X, out=Y), 200 if batch <= 256 else 50)
            res.setdefault((name, batch), []).append(us)
        del g
for (name, batch), v in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, 10240, batch)
    best = min(v)
    print('%-26s batch %5d: %s us per product (best %.1f us = %.1f %% of 8 TB/s)'
          % (name, batch, ' '.join('%8.1f' % x for x in v), best, alg / best / 1e6 / 8000 * 100))
