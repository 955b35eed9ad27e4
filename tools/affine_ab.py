"""A/B of the transform kernels' launch order at a (D, Q, m) shape: the orders of
rounds 1-3 (each kernel's own XCD mapping, 192 MB chunks) against the pair-affine
order of round 4 (rl_kernels2.h: affine_tile; chunks whose intermediates stay in
the XCDs' L2s, two streams).  Structured forms switched off (gate) so that the
transform kernels run whatever the kernel.

    python tools/affine_ab.py [c2|c5|D,Q,m] [batches...]
"""
import os
os.environ.setdefault('RUNLMC_DEBUG', '1')   # the switches below are debug hooks
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd.util import synth          # noqa: E402
from runlmc_amd._native import GridOp      # noqa: E402


def timeit(fn, steps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    if ',' in cfg:
        D, Q, m_data = (int(v) for v in cfg.split(','))
    else:
        D, Q, _, m_data, _ = synth.CONFIGS[cfg]
    batches = [int(b) for b in sys.argv[2:]] or [17, 64, 256, 1024, 4096]
    p = synth.make_problem(D, Q, 1, m_data)
    tops = synth.tops(p)
    variants = [('rounds 1-3 order', {'RUNLMC_AFFINE': '0'}),
                ('pair-affine order', {'RUNLMC_AFFINE': '1'})]
    for kb in os.environ.get('AB_KB', '').split():
        variants.append(('pair-affine, %4s KB/XCD chunks' % kb,
                         {'RUNLMC_AFFINE': '1', 'RUNLMC_AFFINE_KB': kb}))
    if os.environ.get('AB_CONTROLS'):
        # control: the old order in small chunks on two streams (the launch pattern of the
        # L2-sized affine chunks without their placement)
        variants.append(('old order, 20 MB chunks x2 streams',
                         {'RUNLMC_AFFINE': '0', 'RUNLMC_CHUNK_MB': '20', 'RUNLMC_TWO_STREAMS': '1'}))
    ops = []
    for name, env in variants:
        for k in ('RUNLMC_AFFINE', 'RUNLMC_AFFINE_KB', 'RUNLMC_CHUNK_MB', 'RUNLMC_TWO_STREAMS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        g = GridOp(D, p.m, Q)
        g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))
        g.set_form_gate(1 << 62)
        ops.append((name, g))
    print('%s: D=%d Q=%d grid %d L=%d (%d x %d), a pair\'s intermediates %.0f KB'
          % (cfg, D, Q, p.m, ops[0][1].L, ops[0][1].N1, ops[0][1].N2, D * ops[0][1].L * 16 / 1024))
    ref = None
    for b in batches:
        X = torch.randn(b, D * p.m, dtype=torch.float64, device=ops[0][1].device)
        Y = torch.empty_like(X)
        alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, ops[0][1].L, b)
        line = 'batch %5d:' % b
        ref = None
        for name, g in ops:
            us = timeit(lambda: g.mvm(X, out=Y), 200 if b <= 64 else (50 if b <= 1024 else 10))
            if ref is None:
                ref = Y.clone()
            else:
                assert torch.equal(ref, Y), name
            line += '  %s %9.1f us (%.1f %%)' % (name, us, alg / (us * 1e-6) / 8e12 * 100)
        print(line, flush=True)


if __name__ == '__main__':
    main()
