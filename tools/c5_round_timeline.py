"""Timeline of one C5 solver round from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/c5_round_timeline.py run
   python3 tools/c5_round_timeline.py show DIR"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == 'run':
    import numpy as np, torch
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd._native import solve_batch
    D, Q, R, m, npr = synth.CONFIGS['c5']
    p = synth.make_problem(D, Q, R, m)
    fk = synth.functional_kernel(p)
    K, _ = gen_grid_kernel(fk, {(0,): p.grid_dists}, {(0,): (p.W, p.WT)}, p.lens)
    op = K.device_operator()
    rng = np.random.RandomState(3)
    B = torch.from_numpy(rng.randint(0, 2, (npr + 1, p.n)) * 2.0 - 1).to(op.device)
    solve_batch(op, B, tol=1e-4, maxiter=31)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # last complete round: from the last-but-one k_minres2_p to the last one
    ps = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_minres2_p')]
    a, b = ps[-3], ps[-2]
    t0 = int(rows[a]['Start_Timestamp'])
    print('round wall: %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
    busy_end = t0
    idle = 0.0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end, e)
    print('time with no kernel running: %.1f us' % (idle / 1e3))
    names = {}
    for r in rows[a:b]:
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:26]
        d = names.setdefault(n, [0, 0.0, 1e30, 0])
        d[0] += 1; d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        d[2] = min(d[2], (int(r['Start_Timestamp']) - t0) / 1e3); d[3] = max(d[3], (int(r['End_Timestamp']) - t0) / 1e3)
    for n, d in names.items():
        print('%-28s x%3d  sum %8.1f us  first start %8.1f  last end %8.1f' % (n, d[0], d[1], d[2], d[3]))
