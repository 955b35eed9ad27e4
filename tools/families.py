"""The grid product at the reference benchmark's four kernel families
(benchmarks/benchlib/bench.py:94,284-297): forms chosen per top row, time per
product in the chosen forms and on the transform kernels, agreement of the two.

    python tools/families.py [c5|c2] [batch]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from runlmc_amd.util import synth          # noqa: E402
from runlmc_amd._native import GridOp      # noqa: E402


def timeit(fn, steps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
    D, Q, R, m_data, n_probes = synth.CONFIGS[cfg]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else n_probes + 1
    dev = torch.device('cuda', 0)
    print('%s: D=%d Q=%d m=%d batch=%d' % (cfg, D, Q, m_data, batch))
    for kern in synth.KERN_FAMILIES:
        p = synth.make_problem(D, Q, R, m_data, kern=kern)
        g = GridOp(D, p.m, Q)
        t0 = time.perf_counter()
        g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
        forms, structured = g.top_forms()
        torch.cuda.synchronize()
        t_set = (time.perf_counter() - t0) * 1e3
        X = torch.randn(batch, D * p.m, dtype=torch.float64, device=dev)
        Y = torch.empty_like(X)
        Yf = torch.empty_like(X)
        g.set_form_gate(0)
        ms = timeit(lambda: g.mvm(X, out=Y))
        g.set_form_gate(1 << 62)
        msf = timeit(lambda: g.mvm(X, out=Yf))
        g.set_form_gate(-1)
        err = float((Y - Yf).abs().max() / Yf.abs().max())
        alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, batch)
        print('%-9s forms %s structured %s rank %d  set+verify %.2f ms | chosen forms %.3f ms '
              '(%.1f %% of 8 TB/s) | transform kernels %.3f ms (%.1f %%) | max rel diff %.2e'
              % (kern, forms, structured, g.form()[0], t_set, ms, alg / ms / 8e7, msf,
                 alg / msf / 8e7, err), flush=True)


if __name__ == '__main__':
    main()
