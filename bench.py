#!/usr/bin/env python3
"""Headline benchmark: Kronecker-Toeplitz MVMs/sec + NLL-and-gradient
wall-clock at the synthetic (D, Q, m) configurations of BASELINE.json.

    python bench.py --gpus 1 --steps 20 --warmup 5              # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline (`value`): C5 (D=10, Q=5, m=10^5 -> grid 100 004, N=128 probes).  A
step is ONE batched grid product  Y = K_UU X,  K_UU = sum_q B_q (x) T_q,  over
the N+1 = 129 vectors a gradient step's solver iteration carries (y and the
Hutchinson probes this rank owns), inputs resident in HBM before the timed
region.  Ranks hold replicas of the operator and their own probe shard with no
collective in the product (weak scaling; value = all ranks' MVMs / max time).
`value`, `ms_per_step` and `roofline.achieved` come from ONE clock: the wall
clock around the K steps, bracketed by barrier + synchronize (the device-event
time of the same region is printed beside it as a cross-check).

Extra keys of the same JSON line: the full K~ = W K_UU W^T + eps product, the
NLL+gradient step (eps = 0.1 as the reference, and the eps = 1 control of
SURVEY 8d), the same set at C2 under "c2", and `cpu_baseline` -- the oracle
timed on this host in a child process that never touches the GPU (single-core
grid MVMs/sec; NLL+gradient with multiprocessing.Pool(cpu_count()) over the N+1
solves as reference benchmarks/benchlib/bench.py:214-227 does).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')   # reference bench.py:7

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r06', 'traffic.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='c5', choices=['c1', 'c2', 'c5'])
    ap.add_argument('--kern', default='rbf', choices=['rbf', 'periodic', 'matern', 'mix'],
                    help='kernel family of the headline (reference benchmarks/benchlib/'
                         'bench.py:94,284-297); the other three are timed under "families"')
    ap.add_argument('--no-families', action='store_true',
                    help='skip the product timings of the other kernel families')
    ap.add_argument('--batch', type=int, default=0,
                    help='vectors per step (default: probes per GPU + 1)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline')
    ap.add_argument('--no-nll', action='store_true', help='skip the NLL+grad timing')
    ap.add_argument('--no-sweep', action='store_true',
                    help='skip the saturating-batch timings (extra keys)')
    ap.add_argument('--no-full', action='store_true',
                    help='skip the full-operator (W K_UU W^T + eps) timing')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the second configuration (the "c2" key)')
    ap.add_argument('--no-stall', action='store_true',
                    help='skip the C5 step run on to the stall of the residuals (nll_grad_to_stall)')
    ap.add_argument('--stall-iters', type=int, default=3000,
                    help='iteration cap of that step (the C5 residuals stop falling there)')
    ap.add_argument('--cpu-seconds', type=float, default=8.0,
                    help='budget of each bounded CPU sample')
    # debugging aids for the multi-rank path on a one-GPU box
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--force-dist', action='store_true',
                    help='with ONE rank: initialise the process group anyway and push the '
                         "step's broadcast / all-reduce through it (RCCL on a one-GPU box)")
    ap.add_argument('--same-gpu', action='store_true',
                    help='every rank uses cuda:0 (only meaningful with gloo)')
    ap.add_argument('--cpu-full-length', type=int, default=0, metavar='ITERATIONS',
                    help='no GPU: ONE full-length CPU NLL+gradient run of --config (every solve '
                         'this many MINRES iterations, all gradient loops) next to the bounded '
                         "sample's extrapolation; prints a JSON record (profiles/r05/"
                         'cpu_full_length_c5.txt)')
    # internal: the CPU-baseline child process (never imports torch)
    ap.add_argument('--cpu-child', default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------
# CPU baseline: runs in a child process (no GPU runtime in it, so forking a
# process pool is safe), imports oracle/ -- the only part of this file that does
# ---------------------------------------------------------------------------
_POOL_OP = None
_POOL_CAP = 0
_POOL_RULE = False      # MINRES's own stopping tests off: the reference's residual rule ends a solve


def _pool_solve(rhs):
    """One of the N+1 solves of the reference's pool (stochastic_deriv.py:39-52
    -> Iterative.solve): SciPy-rule MINRES + the 100-iteration residual check,
    optionally capped at _POOL_CAP iterations for the bounded C5 sample."""
    from oracle.solver import minres_ps, _EarlyExit
    op = _POOL_OP
    t0 = time.perf_counter()
    ctr = [0]

    def cb(x):
        ctr[0] += 1
        if ctr[0] % 100 == 0 and np.linalg.norm(rhs - op.matvec(x)) < 1e-4:
            raise _EarlyExit(x)
        if _POOL_CAP and ctr[0] >= _POOL_CAP:
            raise _EarlyExit(x)
    try:
        x = minres_ps(op.matvec, rhs, rtol=1e-10, maxiter=len(rhs), callback=cb,
                      own_exits=not _POOL_RULE)[0]
    except _EarlyExit as e:
        x = e.x
    return x, ctr[0], time.perf_counter() - t0


def usable_cores():
    """Cores this process may really use: the affinity mask and the cgroup CPU
    quota, not os.cpu_count() (a container can see 256 CPUs and be allowed 12)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = float(txt[0])
                per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_child(spec_json):
    """Child-process entry: times the oracle on the host cores."""
    global _POOL_OP, _POOL_CAP, _POOL_RULE
    import multiprocessing as mp
    from runlmc_amd.util import synth
    from oracle import likelihood as olik
    from oracle import operators as ops
    from oracle.kernels import KernelSpec, RBFSpec, Matern32Spec, StdPeriodicSpec
    req = json.loads(spec_json)
    out = {}
    cores = usable_cores()
    for name, job in req['jobs'].items():
        D, Q, R, m_data, n_probes = synth.CONFIGS[job['config']]
        p = synth.make_problem(D, Q, R, m_data, eps=job.get('eps', 0.1),
                               kern=job.get('kern', 'rbf'))
        spec = KernelSpec(p.D, synth.kernel_objects(p.kern_desc, rbf=RBFSpec,
                                                    periodic=StdPeriodicSpec,
                                                    matern=Matern32Spec),
                          list(p.coreg_vecs), list(p.coreg_diags), p.noise)
        spec.set_input_dim(1)
        res = {}
        if job.get('mvm'):
            # grid MVMs/sec on ONE core, in the 'sum' representation (what the
            # reference benchmarks force) and the one gen_grid_kernel auto-selects
            x = np.random.RandomState(0).randn(p.D * p.m)
            rates = {}
            for kt in ('sum', olik.choose_ktype(spec)):
                if kt in rates:
                    continue
                op = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens, ktype=kt)
                op.grid_matvec(x)
                op.grid_matvec(x)
                count, t0 = 0, time.perf_counter()
                while True:
                    op.grid_matvec(x)
                    count += 1
                    el = time.perf_counter() - t0
                    if el >= req['seconds'] / 2 or (count >= 2000 and el > 1.0):
                        break
                rates[kt] = count / el
            best = max(rates, key=rates.get)
            res['mvm'] = dict(value=rates[best], representation=best,
                              all={k: round(v, 3) for k, v in rates.items()},
                              seconds_per_representation=req['seconds'] / 2)
        if job.get('nll'):
            # NLL+gradient the reference's way: Pool(cpu_count()) over the N+1
            # solves (bench.py:214-227), then the per-parameter gradient loops in
            # the parent (likelihood.py:48-96)
            np.random.seed(4321)
            probes = np.random.randint(0, 2, (n_probes, p.n)) * 2 - 1
            op = olik.LMCOperatorOracle(spec, p.grid_dists, p.W, p.WT, p.lens)
            _POOL_OP = op
            rhs = [p.y] + [r.astype(np.float64) for r in probes]
            nproc = min(cores, len(rhs))
            _POOL_CAP = 0
            _POOL_RULE = bool(job.get('rule'))
            full_length = int(job.get('full_length_iterations', 0))
            if full_length:
                # every solve runs `full_length` iterations (the device's count), then ALL the
                # gradient loops: the run the bounded sample's extrapolation is checked against
                _POOL_CAP = full_length
            if job.get('bounded'):
                # bounded sample: about 20 s of wall for the pool -- time two
                # operator products here, then cap every solve accordingly
                t0 = time.perf_counter()
                op.matvec(rhs[0])
                op.matvec(rhs[1])
                t_mv = (time.perf_counter() - t0) / 2
                waves = -(-len(rhs) // nproc)
                _POOL_CAP = int(min(60, max(4, 20.0 / (waves * t_mv * 1.5))))
            ctx = mp.get_context('fork')

            def pool_pass():
                t0 = time.perf_counter()
                with ctx.Pool(processes=nproc) as pool:
                    sols_ = pool.map(_pool_solve, rhs, chunksize=1)
                return sols_, time.perf_counter() - t0
            n_params = p.Q * p.R * p.D + p.Q * p.D + p.Q + p.D
            runs = []
            if not _POOL_CAP or full_length:
                # SURVEY 8d / reference grad-grid/slurm-job.sh:52-54: 2 warm-up runs, then
                # >= 5 timed, the MEDIAN reported (every run: the pool over the N+1 solves
                # and the full gradient loops)
                warm, timed = (0, 1) if full_length else (2, 5)
                for it_ in range(warm + timed):
                    sols, solve_wall = pool_pass()
                    t1 = time.perf_counter()
                    olik.stochastic_gradients(spec, p.grid_dists, p.W, p.WT, p.lens, sols[0][0],
                                              probes, np.array([s_[0] for s_ in sols[1:]]))
                    grad_wall = time.perf_counter() - t1
                    if it_ >= warm:
                        runs.append((solve_wall + grad_wall, solve_wall, grad_wall))
                runs.sort()
                _, solve_wall, grad_wall = runs[len(runs) // 2]
            else:
                sols, solve_wall = pool_pass()
            iters = np.array([s[1] for s in sols])
            per_it = float(np.mean([s[2] / max(s[1], 1) for s in sols]))
            alpha = sols[0][0]
            inv_rs = np.array([s[0] for s in sols[1:]])
            info = dict(cores=cores, processes=nproc, rhs=len(rhs),
                        iterations_mean=float(iters.mean()),
                        residual_max=float(max(np.linalg.norm(r - op.matvec(s_[0]))
                                               for r, s_ in zip(rhs[:3], sols[:3]))),
                        solve_wall_s=solve_wall, per_iteration_s=per_it,
                        params=n_params)
            if full_length:
                info.update(grad_wall_s=grad_wall, seconds=solve_wall + grad_wall,
                            iteration_cap=_POOL_CAP, kind='full length, one run',
                            equal_work=True,
                            sample=('Pool(%d) over %d solves of %d MINRES iterations each (the '
                                    "device's count), then all %d parameters x %d right-hand "
                                    'sides of dK products in the parent; ONE run, no warm-up'
                                    % (nproc, len(rhs), _POOL_CAP, n_params, len(rhs))))
            elif not _POOL_CAP:
                info['grad_wall_s'] = grad_wall
                info['seconds'] = solve_wall + grad_wall
                info['seconds_all_runs'] = [round(r[0], 4) for r in runs]
                info['kind'] = 'timed in full'
                info['equal_work'] = True
                info['protocol'] = '2 warm-up runs, 5 timed, median (SURVEY 8d)'
                info['sample'] = ('Pool(%d) over %d solves run to the reference stopping rule, '
                                  'then all %d parameters x %d right-hand sides of dK products '
                                  'in the parent; median of 5 runs after 2 warm-ups'
                                  % (nproc, len(rhs), n_params, len(rhs)))
            else:
                # bounded sample: solves stopped after `cap` iterations and dK
                # products timed on a subset; both scaled linearly (every MINRES
                # iteration and every dK product costs the same)
                T = ops.BTTBOracle(op.tops[0], (p.m,))
                xg = np.random.RandomState(1).randn(p.n)
                reps, t1 = 0, time.perf_counter()
                while time.perf_counter() - t1 < req['seconds'] / 2 or reps < 2:
                    p.W.dot(ops.kron_matvec(op.Bs[0], T, p.WT.dot(xg)))
                    reps += 1
                per_dk = (time.perf_counter() - t1) / reps
                target = float(job['iterations_target'])
                # (the measured wall already spans every wave of the pool)
                solve_est = solve_wall * target / max(float(iters.mean()), 1.0)
                grad_est = per_dk * (n_params - p.D) * len(rhs)
                info.update(per_dK_product_s=per_dk, dK_products_timed=reps,
                            iteration_cap=_POOL_CAP, iterations_target=target,
                            solve_est_s=solve_est, grad_est_s=grad_est,
                            seconds=solve_est + grad_est, kind='extrapolated from a bounded sample',
                            # the pool's solves were STOPPED at the cap (residual_max is theirs,
                            # not that of a finished solve): an iteration-rate sample scaled to
                            # the device's iteration count, not the same result at the same work.
                            # The full-length run it is checked against:
                            # profiles/r05/cpu_full_length_c5.txt
                            equal_work=False,
                            sample=('Pool(%d) over %d solves, each stopped after %d MINRES '
                                    'iterations (measured wall %.1f s) and scaled to the %.0f '
                                    'iterations the device solve took; gradient loops: %d dK '
                                    'products timed on one core, scaled to %d parameters x %d '
                                    'right-hand sides (serial in the parent, as the reference)'
                                    % (nproc, len(rhs), _POOL_CAP, solve_wall, target, reps,
                                       n_params - p.D, len(rhs))))
            res['nll_grad'] = info
        out[name] = res
    out['cpu_model'] = _cpu_model()
    out['cores'] = cores
    out['cpu_count_reported'] = os.cpu_count()
    print('CPUJSON ' + json.dumps(out), flush=True)


def run_cpu_child(jobs, seconds):
    req = json.dumps(dict(jobs=jobs, seconds=seconds))
    env = dict(os.environ, OMP_NUM_THREADS='1', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', req],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
    for line in r.stdout.splitlines():
        if line.startswith('CPUJSON '):
            return json.loads(line[8:])
    raise RuntimeError('CPU baseline child failed (rc %d): %s' % (r.returncode, r.stderr[-2000:]))


# ---------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------
def dist_setup(args):
    import torch
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
        if args.force_dist:
            # a world of ONE rank through the real backend: the step's collectives
            # (alpha broadcast, gradient all-reduce) then run on RCCL itself
            from runlmc_amd.util import dist as rdist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            if args.dist_backend == 'nccl':
                dist.init_process_group('nccl', rank=0, world_size=1,
                                        device_id=torch.device('cuda', 0))
            else:
                dist.init_process_group('gloo', rank=0, world_size=1)
            rdist.force_collectives(True)
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    on = dev if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([x], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def time_steps(fn, steps, warmup, world, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize on
    both sides.  Returns (wall ms per step, device-event ms per step); events
    sit on the stream the library launches on (torch's current stream)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    barrier(world)
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    barrier(world)
    torch.cuda.synchronize(dev)
    wall_ms = (time.perf_counter() - t0) * 1e3
    return wall_ms / steps, e0.elapsed_time(e1) / steps


def measured_traffic(config, batch, form='fft'):
    """HBM-side bytes per step of the grid product from this round's PMC
    passes (tools/traffic_families.py -> TRAFFIC_FILE), or (None, None).
    Entries are keyed config:batch for the transform kernels and
    config:batch:poly for the polynomial form."""
    try:
        table = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    key = '%s:%d' % (config, batch) + ('' if form == 'fft' else ':' + form)
    e = table.get(key)
    return (e['bytes_per_step'], e['source']) if e else (None, None)


def flat_gradient(g):
    """The four gradient families of a step as one vector (coreg vectors, coreg
    diagonals, kernel parameters, noise)."""
    return np.concatenate([np.ravel(x) for x in g[0] + g[1] + [np.hstack(g[2])] + [g[3]]])


def gpu_nll_grad(p, probes_local, n_probes_global, group=None, repeats=2, scipy_exits=True,
                 maxiter=0, keep_gradient=False, precondition=None):
    """One parameters_changed() equivalent on the device: operator update, alpha + probe
    solves, all four gradient families AND the likelihood value (log det K~ + y^T alpha:
    reference models/interpolated_llgp.py:262-290) -- the NLL of "NLL-and-grad".
    precondition=None: the library's default, i.e. the operator's own preconditioner when it
    has one (the Woodbury factorisation of a polynomial-form operator, csrc/rl_direct.h: solves
    end on the reference's residual rule after 1-2 applications, log det exact); False: the
    Krylov solves as until round 5 (scipy_exits=False: run on to the reference's rule,
    RL_MINRES_RULE, at most `maxiter` iterations; log det by Lanczos quadrature).  The
    stopping mode is an argument of the service, not a process-wide switch."""
    import torch
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens,
                             device_index=torch.cuda.current_device())
    svc = StochasticDerivService(None, None, n_probes_global, 1e-4, group=group,
                                 scipy_exits=bool(scipy_exits), maxiter=maxiter,
                                 precondition=precondition)
    best, info = None, None
    # The first pass warms workspaces and graphs.  A FAST step (through the factorisation: tens of
    # milliseconds) gets a second untimed pass and at least three timed ones: the likelihood object
    # of pass k is alive while pass k + 1 allocates its gigabyte of right-hand sides and solutions,
    # so torch's allocator grows for two passes (hipMalloc: +20-30 ms each) before blocks are
    # reused -- a fit sees that once, the steady step is what the line reports
    # (tools/r06_step_times.py: 55, 60, then 28 +- 1 ms over thirty passes; profiles/r06/step_times.txt).
    warm_left, timed, all_s = 1, 0, []
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gks[ad].update(fk, p.grid_dists)
        lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)},
                                  p.Ys, svc, probes=probes_local)
        g = (lik.coreg_vec_gradients(), lik.coreg_diags_gradients(),
             lik.kernel_gradients(), lik.noise_gradient())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        try:
            logdet = lik.log_det_K()
            ll = lik.log_likelihood()
        except ValueError:
            # (solves preconditioned by an INEXACT factorisation -- conjugate gradients, the Matern
            # and mix families: no Lanczos recurrence of K~ ran and the factorisation's log det is
            # not the operator's; the Krylov entry beside this one carries the quadrature)
            logdet = ll = float('nan')
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if warm_left > 0:
            warm_left -= 1
            # (decided from the solver that answered, not from the clock: every rank must run the
            # same number of passes -- they hold collectives)
            if info is None and lik.deriv.lanczos is None:
                warm_left += 1
                repeats = max(repeats, 3)
            info = {}
            continue
        timed += 1
        all_s.append(el)
        if best is None or el < best:
            best = el
            direct = lik.deriv.logdet_exact is not None
            pcg = (not direct) and lik.deriv.lanczos is None
            info = dict(iterations_mean=float(np.mean(lik.deriv.iterations)),
                        iterations_max=int(np.max(lik.deriv.iterations)),
                        residual_max=float(np.max(lik.deriv.residuals)),
                        residual_median=float(np.median(lik.deriv.residuals)),
                        tolerance=1e-4,
                        tolerance_reached=bool(np.max(lik.deriv.residuals) < 1e-4),
                        solver=('direct: K~ = F M F^T + E through the Woodbury identity + '
                                'refinement to the reference\'s residual rule (csrc/rl_direct.h); '
                                'iterations = applications of K~^-1') if direct else
                               ('pcg: conjugate gradients preconditioned by the Woodbury inverse of the '
                                'operator\'s projection on the polynomial subspace (rl_solve_pcg; the '
                                'reference\'s M of iterative.py:47-51), ended by the reference\'s residual rule' +
                                ('; no row of this operator is in the polynomial form: a basis of its own, up '
                                 'to 192 polynomials per output (csrc/rl_solve.hip hz_*)'
                                 if K.device_operator().factor_mode == 3 else ''))
                               if pcg else
                               ('krylov: batched MINRES, ' +
                                ('SciPy 1.15 exits' if scipy_exits else 'reference residual rule only')),
                        nll=float(-ll), logdet=float(logdet),
                        logdet_kind=('exact: determinant lemma on the factorisation' if direct else
                                     ('log det P exactly (determinant lemma on the preconditioner) + preconditioned '
                                      'stochastic Lanczos quadrature of tr log(P^-1/2 K~ P^-1/2): %d extra conjugate-gradient '
                                      'solves of +-1 rows mapped to covariance P (rl_ski_precond_sample, '
                                      'rl_solve_pcg_lanczos)' % K.preconditioner.LOGDET_PROBES) if pcg else
                                     'stochastic Lanczos quadrature from the probe solves'),
                        grad_norm=float(np.linalg.norm(flat_gradient(g))),
                        # a few entries, so that runs can be compared with each other
                        grad_sample=[float(v) for v in np.concatenate(
                            [np.ravel(g[3]), np.ravel(g[1][0]), np.ravel(g[0][0])])[:12]])
            info['seconds_loglik'] = el - (t1 - t0)
            if pcg and lik.deriv.logdet_precond is not None:
                info['logdet_sem'] = float(lik.deriv.logdet_precond[1])
                info['logdet_solves_iterations_max'] = int(np.max(lik.deriv.logdet_precond[2]))
            if not direct and not pcg:
                est = lik.deriv.logdet_probe_estimates()
                if len(est) > 1:
                    info['logdet_sem'] = float(est.std(ddof=1) / np.sqrt(len(est)))
                info['lanczos_steps_kept'] = int(lik.deriv.lanczos.shape[1])
        if timed >= repeats:
            break
    info['seconds_all_timed_passes'] = [float(v) for v in all_s]
    info['seconds_median'] = float(np.median(all_s))
    if group is not None:
        # the same BITS on every rank?  (64-bit checksums of alpha and of the gradient,
        # max and min over the ranks)
        import torch.distributed as tdist
        flat = flat_gradient(g)
        sums = torch.stack([lik.deriv.alpha_dev.view(torch.int64).sum(),
                            torch.from_numpy(flat.copy()).view(torch.int64).sum().to(
                                lik.deriv.alpha_dev.device)])
        on = sums if tdist.get_backend() == 'nccl' else sums.cpu()
        hi, lo = on.clone(), on.clone()
        tdist.all_reduce(hi, op=tdist.ReduceOp.MAX, group=group)
        tdist.all_reduce(lo, op=tdist.ReduceOp.MIN, group=group)
        info['bits_equal_across_ranks'] = {'alpha': bool(hi[0] == lo[0]),
                                           'gradient': bool(hi[1] == lo[1])}
    for k_ in ('nll', 'logdet', 'logdet_sem'):
        if k_ in info and not np.isfinite(info[k_]):
            info[k_] = None                       # (strict JSON has no NaN)
    if keep_gradient:
        info['_gradient'] = flat_gradient(g)      # (popped by the caller: not part of the line)
    info['seconds'] = best
    info['seconds_per_iteration'] = best / max(info['iterations_max'], 1)
    return info


def time_full_operator(g, p, batch, gen, steps, world, dev, alg):
    """K~ = W K_UU W^T + eps on `batch` data-space vectors."""
    import torch
    from runlmc_amd._native import SkiOp
    s = SkiOp(g, p.W, p.WT)
    s.set_noise(p.noise, p.lens)
    Xd = torch.randn(batch, p.n, dtype=torch.float64, generator=gen).to(dev)
    Yd = torch.empty_like(Xd)
    fsteps = max(3, steps // 2)
    fwall, fev = time_steps(lambda: s.mvm(Xd, out=Yd), fsteps, 2, world, dev)
    fwall = max_over_ranks(fwall, world, dev)
    nnz = int(p.W.nnz)
    full_alg = alg + 8 * 3 * p.n * batch + 2 * (nnz * 12 + (p.n + 1) * 4)
    return {'mvm_per_s': batch * world / (fwall * 1e-3), 'ms_per_step': fwall,
            'steps': fsteps, 'n': p.n,
            'roofline_frac': full_alg / (fwall * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'algorithmic_bytes_per_step': full_alg,
            'what': 'K~ x = W (K_UU (W^T x)) + eps * x on %d data-space vectors in the CALLER\'s row '
                    'order.  With every top row in the polynomial form this runs as F M F^T + eps, '
                    'F = W Phi in the caller\'s order (csrc/rl_rowpoly.h: k_rp_project on the fp64 '
                    'matrix cores, k_lr_mix, k_rp_expand; no interpolation product, no grid vector, '
                    'no row permutation of the batch -- until round 4 the two permutations were half '
                    'of this time); other operators: permute, W^T, grid product, W, permute.  The '
                    'algorithmic byte count is that of the interpolation-product algorithm '
                    '(SURVEY 8d); the row-polynomial form moves fewer' % batch}


FORM_NAMES = {0: 'transform', 1: 'polynomial', 2: 'filter'}


def product_form(g, batch, D, m):
    """(label, kernel description) of the form the operator's product runs in
    for this batch: per-top forms from the handle (csrc/rl_lowrank.h,
    rl_filter.h), transform kernels below the batch gate or if any top needs
    them."""
    forms, structured = g.top_forms()
    rank_poly, gate = g.form()
    big = batch * D * m >= gate
    names = [FORM_NAMES[f] for f in forms]
    if (not big and structured and all(f == 1 for f in forms) and D * m <= 32768
            and gate == (1 << 20)):
        # (csrc/rl_lowrank.h: RL_LR_SMALL_MAX; a batch below the gate of an operator wholly in
        # the polynomial form, round 6)
        return 'poly', names, ('grid MVM, polynomial-subspace form for a batch below the gate: two '
                               'launches spread over the chip (k_lr_small_project -> '
                               'k_lr_small_expand, rank %d), D=%d' % (rank_poly, D))
    if not (big and structured):
        return 'fft', names, ('grid MVM (column transforms + row transforms with the D x D mix '
                              '+ adjoint column transforms), D=%d' % D)
    if all(f == 1 for f in forms):
        return 'poly', names, ('grid MVM, polynomial-subspace form (k_lr_project -> k_lr_mix -> '
                               'k_lr_expand, rank %d, accepted at set time against the transform '
                               'kernels), D=%d' % (rank_poly, D))
    if all(f == 2 for f in forms):
        return 'filter', names, ('grid MVM, recursive-filter form of exponential-polynomial top '
                                 'rows (k_sf_carries2 -> k_sf_scan1 -> k_sf_apply: one lane per (row, '
                                 '32-point segment) runs the recurrences, segments chained over DPP '
                                 'rows, persistent workgroups), D=%d' % D)
    return 'filter+poly', names, ('grid MVM, filter part (k_sf_*) + polynomial part (k_lr_*, '
                                  'accumulating), D=%d' % D)


def time_family(kern, name, args, rank, world, dev, steps, warmup):
    """The batched K_UU product of another kernel family of the reference's
    benchmark (bench.py:94,284-297) at the same (D, Q, m) and batch: forms chosen per
    top row, the product in those forms and on the transform kernels, same clock."""
    import torch
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp
    D, Q, R, m_data, n_probes = synth.CONFIGS[name]
    p = synth.make_problem(D, Q, R, m_data, kern=kern)
    g = GridOp(D, p.m, Q, device_index=dev.index)
    g.set_lmc(synth.tops(p), list(p.coreg_vecs), list(p.coreg_diags))
    batch = n_probes + 1
    gen = torch.Generator(device='cpu').manual_seed(1000 + rank)
    X = torch.randn(batch, D * p.m, dtype=torch.float64, generator=gen).to(dev)
    Y = torch.empty_like(X)
    form, names, kernel = product_form(g, batch, D, p.m)
    wall_ms, ev_ms = time_steps(lambda: g.mvm(X, out=Y), steps, warmup, world, dev)
    wall_ms = max_over_ranks(wall_ms, world, dev)
    alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, batch)
    achieved = alg / (wall_ms * 1e-3) / 1e9
    traffic, source = measured_traffic(name, batch, form if kern == 'rbf' else kern)
    out = {'kernels': [list(d) for d in p.kern_desc], 'top_forms': names,
           'mvm_per_s': batch * world / (wall_ms * 1e-3), 'ms_per_step': wall_ms,
           'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                        'traffic_source': source,
                        'traffic_kind': 'recorded' if traffic is not None else None,
                        'kernel': kernel, 'form': form,
                        'algorithmic_bytes_per_step': alg,
                        'device_event_ms_per_step': max_over_ranks(ev_ms, world, dev)}}
    if form != 'fft':
        g.set_form_gate(1 << 62)
        f_ms, _ = time_steps(lambda: g.mvm(X, out=Y), steps, warmup, world, dev)
        g.set_form_gate(-1)
        f_ms = max_over_ranks(f_ms, world, dev)
        out['transform_kernels'] = {'ms_per_step': f_ms,
                                    'roofline_frac': alg / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out


def bench_config(name, args, rank, world, dev, steps, warmup, headline):
    """All device measurements of one (D, Q, m) configuration."""
    import torch
    from runlmc_amd.util import synth
    from runlmc_amd._native import GridOp, SkiOp
    D, Q, R, m_data, n_probes = synth.CONFIGS[name]
    kern = args.kern
    p = synth.make_problem(D, Q, R, m_data, kern=kern)
    g = GridOp(D, p.m, Q, device_index=dev.index)
    tops = synth.tops(p)
    g.set_lmc(tops, list(p.coreg_vecs), list(p.coreg_diags))

    # weak scaling: every rank carries n_probes probes (+ y)
    batch = (args.batch if headline else 0) or (n_probes + 1)
    gen = torch.Generator(device='cpu').manual_seed(1000 + rank)
    X = torch.randn(batch, D * p.m, dtype=torch.float64, generator=gen).to(dev)
    Y = torch.empty_like(X)
    wall_ms, ev_ms = time_steps(lambda: g.mvm(X, out=Y), steps, warmup, world, dev)
    wall_ms = max_over_ranks(wall_ms, world, dev)
    ev_ms = max_over_ranks(ev_ms, world, dev)
    mvms = batch * world / (wall_ms * 1e-3)
    alg = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, batch)
    achieved = alg / (wall_ms * 1e-3) / 1e9
    # which form the product ran in (per-top forms are decided at set time)
    form, form_names, kernel = product_form(g, batch, D, p.m)
    poly = form != 'fft'
    traffic, source = measured_traffic(name, batch, form if kern == 'rbf' else kern)
    out = {
        'value': mvms, 'ms_per_step': wall_ms,
        'config': {'workload': '%s synthetic D=%d Q=%d R=%d m=%d (grid %d, L=%d) '
                               'N=%d probes/GPU, batch=%d vectors/step, K_UU product, '
                               'kernel family %s'
                               % (name, D, Q, R, m_data, p.m, g.L, n_probes, batch, kern),
                   'kern': kern, 'kernels': [list(d) for d in p.kern_desc],
                   'top_forms': form_names,
                   'D': D, 'Q': Q, 'm': p.m, 'L': g.L, 'batch': batch,
                   'fft_split': [g.N1, g.N2], 'parallelism': 'probe-shard x%d' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_source': source,
                     # (read from the committed rocprofv3 --pmc record, not counted in this
                     # run: it cannot notice a regression of the traffic)
                     'traffic_kind': 'recorded' if traffic is not None else None,
                     'kernel': kernel, 'form': form,
                     'algorithmic_bytes_per_step': alg,
                     'clock': 'wall, same region as value',
                     'device_event_ms_per_step': ev_ms},
    }
    if traffic is not None:
        out['roofline']['traffic_GBps'] = traffic / (wall_ms * 1e-3) / 1e9
    if headline:
        # what THIS box does when it merely copies the step's vectors (a stock elementwise
        # kernel: read X once, write Y once -- the algorithmic traffic of the product), same
        # clock: the interface's 8 TB/s is `peak`, this is what a read-once / write-once
        # kernel pair can be held against (tools/lr_pattern_probe.hip)
        c_ms, _ = time_steps(lambda: torch.mul(X, 1.0, out=Y), 20, 5, world, dev)
        c_ms = max_over_ranks(c_ms, world, dev)
        copy_gbs = 2.0 * X.numel() * 8 / (c_ms * 1e-3) / 1e9
        out['roofline']['copy_same_vectors'] = {
            'GBps': copy_gbs, 'ms': c_ms, 'kind': 'measured in this run',
            'what': 'torch.mul(X, 1.0, out=Y) on the step\'s %d x %d fp64 vectors' % tuple(X.shape)}
        out['roofline']['frac_of_copy'] = achieved / copy_gbs

    if poly:
        # the same product forced onto the transform (FFT) kernels, same clock:
        # what Matern / short-length-scale kernels and 2-D grids run
        g.set_form_gate(1 << 62)
        f_ms, f_ev = time_steps(lambda: g.mvm(X, out=Y), steps, warmup, world, dev)
        g.set_form_gate(-1)
        f_ms = max_over_ranks(f_ms, world, dev)
        f_traffic, f_source = measured_traffic(name, batch, 'fft')
        out['transform_kernels'] = {
            'mvm_per_s': batch * world / (f_ms * 1e-3), 'ms_per_step': f_ms,
            'roofline_frac': alg / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'traffic': f_traffic, 'traffic_source': f_source,
            'device_event_ms_per_step': max_over_ranks(f_ev, world, dev),
            'what': 'the same product with the structured forms switched off '
                    '(rl_gridop_set_form_gate): k2_cols_fwd -> k3_rows_mix -> k2_cols_inv'}

    if headline and not args.no_families and world == 1:
        # the reference benchmark's other kernel families at the same shape and batch
        fam = {}
        for other in synth.KERN_FAMILIES:
            if other != kern:
                fam[other] = time_family(other, name, args, rank, world, dev, steps, warmup)
        out['families'] = fam

    if not args.no_full:
        out['full_mvm'] = time_full_operator(g, p, batch, gen, steps, world, dev, alg)

    if not args.no_sweep and world == 1:
        # the N+1-vector batch is what a solver round carries; larger batches
        # show what the same kernels sustain
        sweep = {}
        for b in (64, 256, 1024, 4096):
            if b == batch or b * D * p.m * 8 * 2 > 4.2e9:
                continue
            Xb = torch.randn(b, D * p.m, dtype=torch.float64, device=dev)
            Yb = torch.empty_like(Xb)
            w_ms, _ = time_steps(lambda: g.mvm(Xb, out=Yb), max(3, steps // 4), 2, 1, dev)
            ab = synth.algorithmic_bytes_grid_mvm(D, Q, p.m, g.L, b)
            sweep[str(b)] = {'mvm_per_s': b / (w_ms * 1e-3),
                             'roofline_frac': ab / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del Xb, Yb
        out['batch_sweep'] = sweep
    del X, Y

    if not args.no_nll:
        import torch.distributed as tdist
        group = tdist.group.WORLD if (args.force_dist or world > 1) else None

        def rel_dist(a, b):
            return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        for key, eps in (('nll_grad', 0.1), ('nll_grad_eps1', 1.0)):
            pe = p if eps == 0.1 else synth.make_problem(D, Q, R, m_data, eps=eps, kern=kern)
            np.random.seed(4321)
            # the config's N probes in total, dealt round-robin to the ranks
            # (strong scaling of one optimiser step)
            # (drawn as the reference draws them and as StochasticDerivService.draw_probes
            # hands them to generate(): int64.  Checking and narrowing them to one byte per
            # entry is INSIDE the timed step, as it is for a caller of the model)
            probes = np.random.randint(0, 2, (n_probes, pe.n)) * 2 - 1
            # THE step: the library's default path -- the operator's own preconditioner when it
            # has one (rbf, periodic: solves through the factorisation, below the reference's
            # 1e-4 rule; log det exact), Krylov solves otherwise (matern, mix)
            info = gpu_nll_grad(pe, probes, n_probes, group=group, keep_gradient=True)
            grad_default = info.pop('_gradient')
            info['seconds'] = max_over_ranks(info['seconds'], world, dev)
            info.update(n_probes_global=n_probes, scaling='strong', eps=eps,
                        probes_per_rank=-(-n_probes // world),
                        probes='int64 host matrix as the reference draws it, checked and '
                               'narrowed inside the timed step')
            direct = info['solver'].startswith(('direct', 'pcg'))
            if world > 1 and key == 'nll_grad':
                # the SAME step on one rank alone (every rank runs it, no collective): what the
                # probe sharding of this run is a speed-up of (reference axis:
                # lmc/stochastic_deriv.py:39-52)
                one = gpu_nll_grad(pe, probes, n_probes, group=None, repeats=1)
                one_s = max_over_ranks(one['seconds'], world, dev)
                info['one_rank_seconds'] = one_s
                info['speedup_vs_one_rank'] = one_s / info['seconds']
            if world == 1 and key == 'nll_grad':
                # the same step with the probes already ON the device (int8, what
                # StochasticDerivService.draw_probes_device returns): no host pass over N x n
                pd = torch.from_numpy(probes.astype(np.int8)).to(dev)
                dv = gpu_nll_grad(pe, pd, n_probes, group=None)
                info['probes_on_device'] = {'seconds': dv['seconds'], 'nll': dv['nll'],
                                            'residual_max': dv['residual_max']}
                del pd
            if world == 1 and key == 'nll_grad' and n_probes >= 16 and not args.no_extra:
                # one rank's share of an 8-way probe split (N / 8 probes + y) timed on THIS
                # GPU: what the probe sharding can give at most on 8 GPUs -- a projection
                # (no collective, no second GPU involved), not a measurement of scaling
                share = n_probes // 8
                sh = gpu_nll_grad(pe, probes[:share], share, group=None, repeats=2)
                info['projected_strong_scaling_8gpu'] = {
                    'kind': 'projection from one GPU', 'probes_per_rank': share,
                    'seconds_full': info['seconds'], 'seconds_share': sh['seconds'],
                    'ceiling': info['seconds'] / sh['seconds'],
                    'iterations_max_share': sh['iterations_max']}
            out[key] = info
            if key != 'nll_grad' or not direct:
                continue
            # the reference's rule is met: say so under the key the review asked for
            out['nll_grad_to_tolerance'] = {
                k_: info[k_] for k_ in ('seconds', 'tolerance', 'tolerance_reached', 'residual_max',
                                        'residual_median', 'iterations_max', 'nll', 'logdet',
                                        'logdet_kind', 'solver')}
            # ... and the Krylov step of rounds 1-5 next to it (SciPy 1.15's exits: what the
            # reference does today and what the CPU baseline below runs), with its Lanczos
            # log det against the exact one
            kry = gpu_nll_grad(pe, probes, n_probes, group=group, keep_gradient=True,
                               precondition=False)
            grad_kry = kry.pop('_gradient')
            kry['seconds'] = max_over_ranks(kry['seconds'], world, dev)
            if world > 1:
                # the probe-parallel axis proper (reference lmc/stochastic_deriv.py:39-52): the
                # same Krylov step on one rank alone against the sharded one
                one = gpu_nll_grad(pe, probes, n_probes, group=None, repeats=1, precondition=False)
                one_s = max_over_ranks(one['seconds'], world, dev)
                kry['one_rank_seconds'] = one_s
                kry['speedup_vs_one_rank'] = one_s / kry['seconds']
            kry.update(n_probes_global=n_probes, eps=eps,
                       gradient_rel_distance_to_default_step=rel_dist(grad_kry, grad_default),
                       logdet_exact=info['logdet'],            # (exact, or the preconditioned quadrature's)
                       logdet_default_step_kind=info.get('logdet_kind'),
                       logdet_slq_minus_exact=(kry['logdet'] - info['logdet']
                                               if kry['logdet'] is not None and info['logdet'] is not None
                                               else None))
            if kry.get('logdet_sem') and kry['logdet'] is not None and info['logdet'] is not None:
                kry['logdet_slq_minus_exact_in_sem'] = (kry['logdet'] - info['logdet']) / kry['logdet_sem']
            if world == 1 and n_probes >= 16 and not args.no_extra:
                share = n_probes // 8
                sh = gpu_nll_grad(pe, probes[:share], share, group=None, repeats=2,
                                  precondition=False)
                kry['projected_strong_scaling_8gpu'] = {
                    'kind': 'projection from one GPU', 'probes_per_rank': share,
                    'seconds_full': kry['seconds'], 'seconds_share': sh['seconds'],
                    'ceiling': kry['seconds'] / sh['seconds'],
                    'iterations_max_share': sh['iterations_max']}
            out['nll_grad_krylov'] = kry
            if name == 'c5' and world == 1 and not args.no_stall:
                # The Krylov solve run ON: MINRES's own tests off (RL_MINRES_RULE), the
                # reference's rule (explicit residual < 1e-4 every 100 iterations,
                # approx/iterative.py:36-42) until the fp64 residuals stall -- no C5 system
                # reaches 1e-4 that way, they stop falling at ~3e-3 from 3000 iterations on
                # (profiles/r04/time_to_tolerance_c5.txt).  Its gradient against the step
                # that DID reach the tolerance (the default one above).
                st = gpu_nll_grad(pe, probes, n_probes, group=None, repeats=1,
                                  scipy_exits=False, maxiter=args.stall_iters,
                                  keep_gradient=True, precondition=False)
                grad_stall = st.pop('_gradient')
                st.update(n_probes_global=n_probes, eps=eps, maxiter=args.stall_iters,
                          stopping='reference residual rule only (RL_MINRES_RULE), capped at '
                                   'the stall of the fp64 residuals',
                          gradient_rel_distance_to_default_step=rel_dist(grad_stall, grad_default),
                          gradient_rel_distance_to_scipy_exit_step=rel_dist(grad_kry, grad_stall))
                out['nll_grad_to_stall'] = st
        if name != 'c5' and world == 1:
            # the Krylov step run ON to the reference's tolerance: MINRES's own stopping tests
            # off (RL_MINRES_RULE), the reference's rule -- explicit residual < 1e-4 at every
            # 100th iteration, approx/iterative.py:36-42 -- ends each system; the CPU side runs
            # the same (cpu_baseline of this config, job ':rule': equal work)
            np.random.seed(4321)
            probes = np.random.randint(0, 2, (n_probes, p.n)) * 2 - 1
            info = gpu_nll_grad(p, probes, n_probes, group=None, scipy_exits=False,
                                precondition=False)
            info.update(n_probes_global=n_probes, eps=0.1, stopping='reference residual rule only')
            out['nll_grad_krylov_to_tolerance'] = info
    return out


def main():
    args = parse()
    if args.cpu_child is not None:
        cpu_child(args.cpu_child)
        return
    if args.cpu_full_length:
        it = args.cpu_full_length
        jobs = {'full': dict(config=args.config, nll=True, kern=args.kern,
                             full_length_iterations=it),
                'bounded': dict(config=args.config, nll=True, kern=args.kern, bounded=True,
                                iterations_target=float(it))}
        cpu = run_cpu_child(jobs, args.cpu_seconds)
        full, est = cpu['full']['nll_grad'], cpu['bounded']['nll_grad']
        print(json.dumps({'config': args.config, 'kern': args.kern, 'iterations': it,
                          'cpu_model': cpu['cpu_model'], 'cores': cpu['cores'],
                          'host_cpu_count': cpu['cpu_count_reported'],
                          'full_length': full, 'bounded_sample': est,
                          'extrapolation_over_full_length': est['seconds'] / full['seconds']}))
        return
    # (before anything touches the GPU) one rank per GPU: --gpus N needs N ranks.  Started
    # without a launcher (`python bench.py --gpus N`), this process starts them itself -- N fresh
    # children through torch.distributed.run, as a SUBPROCESS whose output and status it relays
    # (never an exec: nothing here has touched the GPU yet, and nothing will in this process).
    world_env = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and world_env is None:
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        raise SystemExit(subprocess.call(cmd, env=env))
    world_env = int(world_env or '1')
    if args.gpus != world_env:
        raise SystemExit(
            'bench.py --gpus %d but WORLD_SIZE=%d: launch N ranks with\n  python -m '
            'torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 '
            '--master-port 29500 bench.py --gpus %d ...   (or plain `python bench.py --gpus %d`, '
            'which starts them itself)'
            % (args.gpus, world_env, args.gpus, args.gpus, args.gpus))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: runlmc_amd has no CPU path')
    rank, world, local = dist_setup(args)
    dev = torch.device('cuda', torch.cuda.current_device())

    head = bench_config(args.config, args, rank, world, dev, args.steps, args.warmup, True)
    out = {
        'metric': 'kronecker_toeplitz_mvms_per_sec',
        'value': head.pop('value'), 'unit': 'MVM/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': head.pop('ms_per_step'),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
    }
    out.update(head)
    other = 'c2' if args.config != 'c2' else None
    if other and not args.no_extra and world == 1:
        # the small synthetic configuration of BASELINE.json as an extra key
        # (17 vectors per step: launch-latency-bound by construction)
        out[other] = bench_config(other, args, rank, world, dev, max(args.steps, 100),
                                  max(args.warmup, 10), False)

    if rank == 0 and world == 1 and not args.no_cpu:
        jobs = {args.config: dict(config=args.config, mvm=True, kern=args.kern)}
        if 'nll_grad' in out:
            # (the CPU side runs the reference's Krylov solves: its sample is scaled to the
            # device's KRYLOV step, not to the one or two applications of the factorisation)
            it = out.get('nll_grad_krylov', out['nll_grad'])['iterations_mean']
            small = args.config != 'c5'
            jobs[args.config].update(
                nll=True, **({} if small else
                             dict(bounded=True, iterations_target=it)))
        if other in out:
            jobs[other] = dict(config=other, mvm=True, nll='nll_grad' in out[other], kern=args.kern)
            if 'nll_grad_krylov_to_tolerance' in out[other]:
                jobs[other + ':rule'] = dict(config=other, nll=True, rule=True, kern=args.kern)
        cpu = run_cpu_child(jobs, args.cpu_seconds)
        mine = cpu[args.config]
        out['cpu_baseline'] = dict(
            value=mine['mvm']['value'], unit='MVM/s', cores=1, kind='port',
            sample='%.0f s of single-vector grid MVMs per representation (oracle, NumPy '
                   'pocketfft, OMP_NUM_THREADS=1, one core); representation=%s; all=%s'
                   % (mine['mvm']['seconds_per_representation'], mine['mvm']['representation'],
                      mine['mvm']['all']),
            cpu_model=cpu['cpu_model'], host_cores_usable=cpu['cores'],
            host_cpu_count=cpu['cpu_count_reported'])
        out['speedup_vs_cpu_mvm'] = out['value'] / mine['mvm']['value']
        if 'nll_grad' in mine:
            out['cpu_baseline']['nll_grad'] = mine['nll_grad']
            out['nll_grad']['speedup_vs_cpu'] = mine['nll_grad']['seconds'] / out['nll_grad']['seconds']
            out['nll_grad']['cpu_kind'] = mine['nll_grad']['kind']
            out['nll_grad']['cpu_seconds_per_iteration_per_solve'] = mine['nll_grad']['per_iteration_s']
            if 'nll_grad_krylov' in out:
                # same algorithm on both sides (the reference's MINRES with SciPy's exits) ...
                k = out['nll_grad_krylov']
                k['speedup_vs_cpu'] = mine['nll_grad']['seconds'] / k['seconds']
                k['cpu_kind'] = mine['nll_grad']['kind']
                k['cpu_equal_work'] = bool(mine['nll_grad'].get('equal_work', False))
                # ... while the default step does MORE than the CPU run it is divided into: it
                # meets the reference's tolerance, the CPU's solves stop where SciPy stops
                out['nll_grad']['cpu_equal_work'] = False
                out['nll_grad']['cpu_note'] = ('the CPU run is the reference\'s Krylov step (residuals '
                                               'above the tolerance at its exit); this step reaches '
                                               'the tolerance')
            else:
                out['nll_grad']['cpu_equal_work'] = bool(mine['nll_grad'].get('equal_work', False))
        if other in cpu:
            o = cpu[other]
            out[other]['cpu_baseline'] = dict(
                value=o['mvm']['value'], unit='MVM/s', cores=1, kind='port',
                sample='as the headline; representation=%s; all=%s'
                       % (o['mvm']['representation'], o['mvm']['all']))
            out[other]['speedup_vs_cpu_mvm'] = out[other]['value'] / o['mvm']['value']
            if 'nll_grad' in o:
                out[other]['cpu_baseline']['nll_grad'] = o['nll_grad']
                out[other]['nll_grad']['speedup_vs_cpu'] = \
                    o['nll_grad']['seconds'] / out[other]['nll_grad']['seconds']
                out[other]['nll_grad']['cpu_kind'] = o['nll_grad']['kind']
            if other + ':rule' in cpu:
                r = cpu[other + ':rule']['nll_grad']
                t = out[other]['nll_grad_krylov_to_tolerance']
                t['cpu'] = r
                t['speedup_vs_cpu'] = r['seconds'] / t['seconds']

    if world > 1 or args.force_dist:
        import torch.distributed as dist
        out['collectives'] = {'backend': dist.get_backend(), 'world_size': world,
                              'forced_in_world_of_one': bool(args.force_dist and world == 1),
                              'per_step': 'broadcast of alpha (n doubles) + one all-reduce of '
                                          'the gradient partials'}
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
