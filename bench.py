#!/usr/bin/env python3
"""Headline benchmark: Kronecker-Toeplitz MVMs/sec (+ NLL-and-gradient
wall-clock) at a synthetic (D, Q, m) configuration of BASELINE.json.

    python bench.py --gpus 1 --steps 50 --warmup 5              # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is ONE batched grid product  Y = K_UU X,  K_UU = sum_q B_q (x) T_q, over
the N+1 vectors a gradient step's solver iteration carries (y and the N
Hutchinson probes this rank owns).  Inputs are resident in HBM before the timed
region.  Ranks hold replicas of the operator and their own probe shard, with
no collective in the product (weak scaling; value = all ranks' MVMs / time).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')   # reference bench.py:7

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from runlmc_amd.util import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# HBM-side bytes per step from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in
# separate runs; FETCH_SIZE of the 16-B/lane reads of the intermediates doubled
# per the gfx950 correction of MI355X_MICROARCH.md; Infinity-Cache hits are
# counted by these counters).  Keyed by (config, batch); source file alongside.
MEASURED_TRAFFIC = {
    # FETCH_SIZE + WRITE_SIZE of the three kernels per step; FETCH_SIZE of the
    # 16-byte-per-lane reads of T doubled (gfx950 correction, MI355X_MICROARCH.md)
    ('c2', 17): (29.7e6, 'profiles/r01/v5_c2_k17_pmc_summary.txt'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c2', choices=sorted(synth.CONFIGS))
    ap.add_argument('--batch', type=int, default=0,
                    help='vectors per step (default: probes per GPU + 1)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline')
    ap.add_argument('--no-nll', action='store_true', help='skip the NLL+grad timing')
    ap.add_argument('--no-sweep', action='store_true',
                    help='skip the saturating-batch timings (extra keys)')
    ap.add_argument('--cpu-seconds', type=float, default=8.0)
    # debugging aids for the multi-rank path on a one-GPU box
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--same-gpu', action='store_true',
                    help='every rank uses cuda:0 (only meaningful with gloo)')
    return ap.parse_args()


def dist_setup(args):
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if args.same_gpu else int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    on = dev if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([x], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def time_grid_mvm(gridop, X, Y, steps, warmup, world):
    """Events on the stream the library launches on (torch's current)."""
    dev = X.device
    for _ in range(warmup):
        gridop.mvm(X, out=Y)
    torch.cuda.synchronize(dev)
    barrier(world)
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        gridop.mvm(X, out=Y)
    e1.record()
    torch.cuda.synchronize(dev)
    barrier(world)
    torch.cuda.synchronize(dev)
    wall_ms = (time.perf_counter() - t0) * 1e3
    ev_ms = e0.elapsed_time(e1)
    return wall_ms / steps, ev_ms / steps


def cpu_baseline(p, tops, seconds):
    """The oracle (NumPy restatement of the reference path) on ONE host core:
    grid MVMs/sec in the 'sum' representation (what BASELINE.json's metric
    names; reference benchmarks force it with ktype='sum') and in the
    representation gen_grid_kernel would auto-select."""
    from oracle import operators as ops
    from oracle import likelihood as olik
    from oracle.kernels import KernelSpec, RBFSpec
    spec = KernelSpec(p.D, [RBFSpec(g) for g in p.inv_lengthscales],
                      list(p.coreg_vecs), list(p.coreg_diags), p.noise)
    spec.set_input_dim(1)
    rng = np.random.RandomState(0)
    x = rng.randn(p.D * p.m)
    out = {}
    for kt in ('sum', olik.choose_ktype(spec)):
        if kt in out:
            continue
        op = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens, ktype=kt)
        op.grid_matvec(x)                       # warm-up
        op.grid_matvec(x)
        count, t0 = 0, time.perf_counter()
        budget = seconds / 2
        while True:
            op.grid_matvec(x)
            count += 1
            el = time.perf_counter() - t0
            if el >= budget or (count >= 2000 and el > 1.0):
                break
        out[kt] = count / el
    best = max(out, key=out.get)
    return dict(value=out[best], unit='MVM/s', cores=1, kind='port',
                sample='%.0f s of single-vector grid MVMs per representation '
                       '(oracle, NumPy pocketfft, OMP_NUM_THREADS=1); '
                       'representation=%s; all=%s'
                       % (seconds / 2, best,
                          {k: round(v, 2) for k, v in out.items()}),
                cpu_model=_cpu_model()), spec


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_nll_grad(p, spec, probes, iters_hint, seconds):
    """Reference-path NLL+gradient wall-clock on the host, bounded: times a
    few MINRES iterations and gradient MVMs of the oracle and scales by the
    counts the full step needs when the full step would not fit the budget."""
    from oracle import likelihood as olik
    from oracle.solver import minres_ps
    op = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
    n_rhs = len(probes) + 1
    cores = os.cpu_count() or 1
    # time MINRES iterations on y
    its = 0
    t0 = time.perf_counter()

    class _Stop(Exception):
        pass

    def cb(_x):
        nonlocal its
        its += 1
        if time.perf_counter() - t0 > seconds / 2 or its >= iters_hint:
            raise _Stop()
    try:
        minres_ps(op.matvec, p.y, rtol=1e-10, maxiter=p.n, callback=cb)
    except _Stop:
        pass
    per_it = (time.perf_counter() - t0) / max(its, 1)
    # gradient side: P (N+1) single-term operator products (likelihood.py:48-96)
    from oracle import operators as ops
    T = ops.BTTBOracle(op.tops[0], (p.m,))
    t1 = time.perf_counter()
    reps = 0
    xg = np.random.RandomState(1).randn(p.n)
    while time.perf_counter() - t1 < seconds / 4 or reps < 2:
        p.W.dot(ops.kron_matvec(op.Bs[0], T, p.WT.dot(xg)))
        reps += 1
    per_dk = (time.perf_counter() - t1) / reps
    n_params = p.Q * p.R * p.D + p.Q * p.D + p.Q     # A_q, kappa_q, one RBF param each
    solve_serial = per_it * iters_hint * n_rhs
    grad_serial = per_dk * n_params * n_rhs
    est = solve_serial / min(cores, n_rhs) + grad_serial
    return dict(est_seconds=est, per_iteration_s=per_it, per_dK_mvm_s=per_dk,
                iterations_assumed=int(iters_hint), rhs=n_rhs, params=n_params,
                cores=cores,
                sample='%d MINRES iterations on y + %d single-term dK products '
                       'timed on one core; scaled to %d rhs x %d iterations '
                       '(solves spread over min(cores, rhs) processes as the '
                       'reference pool does) + %d params x %d rhs dK products'
                       % (its, reps, n_rhs, iters_hint, n_params, n_rhs))


def gpu_nll_grad(p, probes_local, n_probes_global, group=None, repeats=3):
    """One parameters_changed() equivalent on the device: operator update,
    alpha + probe solves, all four gradient families."""
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens,
                             device_index=torch.cuda.current_device())
    svc = StochasticDerivService(None, None, n_probes_global, 1e-4, group=group)
    best, info = None, None
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gks[ad].update(fk, p.grid_dists)
        lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)},
                                  p.Ys, svc, probes=probes_local)
        g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(),
             lik.kernel_gradients(), lik.noise_gradient())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if best is None or el < best:
            best = el
            info = dict(iterations_mean=float(np.mean(lik.deriv.iterations)),
                        iterations_max=int(np.max(lik.deriv.iterations)),
                        residual_max=float(np.max(lik.deriv.residuals)),
                        grad_norm=float(np.sqrt(sum(np.sum(np.square(x)) for x in
                                                    g[0] + g[1] + [np.array(g[2])] + [g[3]]))))
    info['seconds'] = best
    return info


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: runlmc_amd has no CPU path')
    rank, world, local = dist_setup(args)
    dev = torch.device('cuda', torch.cuda.current_device())
    D, Q, R, m_data, n_probes = synth.CONFIGS[args.config]
    p = synth.make_problem(D, Q, R, m_data)
    from runlmc_amd._native import GridOp
    g = GridOp(D, p.m, Q, device_index=dev.index)
    tops = synth.tops(p)
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))

    # weak scaling: every rank carries n_probes probes (+ y)
    batch = args.batch or (n_probes + 1)
    gen = torch.Generator(device='cpu').manual_seed(1000 + rank)
    X = torch.randn(batch, D * p.m, dtype=torch.float64, generator=gen).to(dev)
    Y = torch.empty_like(X)
    wall_ms, ev_ms = time_grid_mvm(g, X, Y, args.steps, args.warmup, world)
    wall_ms = max_over_ranks(wall_ms, world, dev)
    ev_ms = max_over_ranks(ev_ms, world, dev)
    mvms = batch * world / (wall_ms * 1e-3)
    alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, batch)
    achieved = alg / (ev_ms * 1e-3) / 1e9

    out = {
        'metric': 'kronecker_toeplitz_mvms_per_sec',
        'value': mvms, 'unit': 'MVM/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': wall_ms, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s synthetic D=%d Q=%d R=%d m=%d (grid %d, L=%d) '
                               'N=%d probes/GPU, batch=%d vectors/step'
                               % (args.config, D, Q, R, m_data, p.m, g.L, n_probes, batch),
                   'D': D, 'Q': Q, 'm': p.m, 'L': g.L, 'batch': batch,
                   'fft_split': [g.N1, g.N2], 'parallelism': 'probe-shard x%d' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': MEASURED_TRAFFIC.get((args.config, batch), (None, None))[0],
                     'traffic_source': MEASURED_TRAFFIC.get((args.config, batch), (None, None))[1],
                     'kernel': 'grid MVM = k2_cols_fwd + k2_rows_mix<%d> + k2_cols_inv' % D,
                     'algorithmic_bytes_per_step': alg,
                     'device_ms_per_step': ev_ms},
    }
    tr = out['roofline']['traffic']
    if tr is not None and ev_ms > 0:
        # rate at which the MEASURED bytes move (the two-level transform moves
        # ~5x the algorithmic bytes: DESIGN.md section 5)
        out['roofline']['traffic_GBps'] = tr / (ev_ms * 1e-3) / 1e9

    if not args.no_sweep and world == 1:
        # the configured batch (N+1 vectors) is latency-bound on a chip this
        # size; larger batches show what the same kernels sustain
        sweep = {}
        for b in (64, 256, 1024, 4096):
            if b * D * p.m * 8 * 2 > 4e9:
                continue
            Xb = torch.randn(b, D * p.m, dtype=torch.float64, device=dev)
            Yb = torch.empty_like(Xb)
            _, ms = time_grid_mvm(g, Xb, Yb, max(3, args.steps // 10), 2, 1)
            ab = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, b)
            sweep[str(b)] = {'mvm_per_s': b / (ms * 1e-3),
                             'roofline_frac': ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'path': 'on-chip' if g.onchip[0] and b >= g.onchip[1]
                             else 'three-kernel'}
        out['batch_sweep'] = sweep

    if not args.no_nll:
        import torch.distributed as dist
        np.random.seed(4321)
        # the config's N probes in total, dealt round-robin to the ranks
        # (strong scaling of one optimiser step; alpha is solved on every rank)
        total = n_probes
        probes = np.random.randint(0, 2, (total, p.n)) * 2 - 1
        info = gpu_nll_grad(p, probes, total)
        info['seconds'] = max_over_ranks(info['seconds'], world, dev)
        info['n_probes_global'] = total
        info['scaling'] = 'strong'
        info['probes_per_rank'] = -(-total // world)
        out['nll_grad'] = info

    if rank == 0 and world == 1 and not args.no_cpu:
        base, spec = cpu_baseline(p, tops, args.cpu_seconds)
        out['cpu_baseline'] = base
        out['speedup_vs_cpu_mvm'] = mvms / base['value']
        if 'nll_grad' in out:
            hint = out['nll_grad']['iterations_mean']
            cpu = cpu_nll_grad(p, spec, probes, max(int(round(hint)), 1),
                               args.cpu_seconds)
            out['cpu_baseline']['nll_grad'] = cpu
            out['nll_grad']['speedup_vs_cpu_est'] = cpu['est_seconds'] / out['nll_grad']['seconds']

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
