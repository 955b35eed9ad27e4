"""Probe sharding over ranks (the N > 1 path): two gloo processes on CPU, the
kernels running under the emulator, must reproduce the one-process gradients:
probes dealt round-robin, alpha solved on every rank, one all-reduce of the
partial sums."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _grads(world_group=None):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import parity_suite as ps
    from cases import Case
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    c = Case('lmc_small')
    fk, K, gk = ps.build_operator(c)
    ad = (0,)
    svc = StochasticDerivService(None, None, len(c.rs), 1e-4, group=world_group)
    lik = ApproxLMCLikelihood(fk, K, {ad: c.grid_dists}, {ad: (c.W, c.WT)},
                              c.Ys, svc, probes=c.rs)
    flat = np.concatenate([np.ravel(g) for g in lik.coreg_vec_gradients()] +
                          [np.ravel(g) for g in lik.coreg_diags_gradients()] +
                          [np.ravel(g) for g in lik.kernel_gradients()] +
                          [lik.noise_gradient()])
    return flat, lik.deriv.rs_dev.shape[0], lik.deriv.alpha.copy()


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    flat, nloc, alpha = _grads()
    q.put((rank, flat, nloc, alpha))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_probe_sharding():
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    try:
        ref, nall, alpha_ref = _grads()
    finally:
        _lib.use_library(None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][2] + got[1][2] == nall          # every probe owned once
    assert got[0][2] == (nall + 1) // 2
    # alpha is solved on every rank in a transform pair of its own (one rank
    # holds an odd, the other an even number of probes: both batch layouts):
    # the SAME BITS everywhere, with no broadcast -- and the same as one rank's
    assert np.array_equal(got[0][3], got[1][3])
    assert np.array_equal(got[0][3], alpha_ref)
    # both ranks end with the same, full gradient (identical alpha terms,
    # identical all-reduced probe sums)
    assert np.array_equal(got[0][1], got[1][1])
    # batching changes which vectors share a transform -> solver-level noise
    scale = np.abs(ref).max()
    assert np.abs(got[0][1] - ref).max() < 1e-4 * scale


def _grads_smooth(world_group=None):
    """The step on the golden case whose top rows are all smooth: solves through the operator's
    factorisation (csrc/rl_direct.h), Gram terms in coefficient space, exact log det."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import parity_suite as ps
    from cases import Case
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    c = Case('lmc_smooth')
    fk, K, gk = ps.build_operator(c)
    ad = (0,)
    svc = StochasticDerivService(None, None, len(c.rs), 1e-9, group=world_group)
    lik = ApproxLMCLikelihood(fk, K, {ad: c.grid_dists}, {ad: (c.W, c.WT)},
                              c.Ys, svc, probes=c.rs)
    assert lik.deriv.logdet_exact is not None          # the direct path answered
    flat = np.concatenate([np.ravel(g) for g in lik.coreg_vec_gradients()] +
                          [np.ravel(g) for g in lik.coreg_diags_gradients()] +
                          [np.ravel(g) for g in lik.kernel_gradients()] +
                          [lik.noise_gradient()])
    return flat, lik.deriv.rs_dev.shape[0], lik.deriv.alpha.copy(), lik.log_likelihood(), c


def _worker_smooth(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    flat, nloc, alpha, ll, _ = _grads_smooth()
    q.put((rank, flat, nloc, alpha, ll))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_direct_path():
    """Probe sharding when the solves go through the operator's factorisation (round 6): every
    rank factors its replica, solves alpha and its share of the probes, rank 0's alpha is
    broadcast, ONE all-reduce carries the Gram terms -- the same bits on both ranks, the
    one-rank step's gradient to 1e-9 (the solves are converged: no solver-level noise), the
    reference's dense values (golden lmc_smooth) to 1e-8, the same exact log likelihood."""
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    try:
        ref, nall, alpha_ref, ll_ref, c = _grads_smooth()
    finally:
        _lib.use_library(None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_smooth, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][2] + got[1][2] == nall
    assert np.array_equal(got[0][3], got[1][3])            # alpha: rank 0's, broadcast
    assert np.array_equal(got[0][1], got[1][1])            # the assembled gradient
    assert got[0][4] == got[1][4]
    scale = np.abs(ref).max()
    assert np.abs(got[0][1] - ref).max() < 1e-9 * scale
    assert np.abs(got[0][3] - c.g['alpha_dense']).max() < 1e-8 * np.abs(c.g['alpha_dense']).max()
    assert abs(got[0][4] - ll_ref) <= 1e-12 * abs(ll_ref)


def _grads_matern(world_group=None):
    """A step on a synthetic problem whose rows are all Matern (filter form): the solves run
    conjugate gradients preconditioned on the larger basis (RUNLMC_PRECOND_HI_MIN lowered)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ['RUNLMC_DEBUG'] = '1'
    os.environ['RUNLMC_PRECOND_HI_MIN'] = '0'
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    p = synth.make_problem(2, 2, 1, 1000, kern='matern')
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, _ = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    assert K.device_operator().factor()[0] and K.device_operator().factor_mode == 3
    rs = np.random.RandomState(5).randint(0, 2, (6, p.n)) * 2 - 1
    svc = StochasticDerivService(None, None, len(rs), 1e-9, group=world_group)
    lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys, svc, probes=rs)
    assert lik.deriv.lanczos is None and lik.deriv.logdet_exact is None   # the preconditioned path answered
    flat = np.concatenate([np.ravel(g) for g in lik.coreg_vec_gradients()] +
                          [np.ravel(g) for g in lik.coreg_diags_gradients()] +
                          [np.ravel(g) for g in lik.kernel_gradients()] +
                          [lik.noise_gradient()])
    return flat, lik.deriv.rs_dev.shape[0], lik.deriv.alpha.copy(), lik.log_det_K(), lik.deriv.logdet_precond[1]


def _worker_matern(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    flat, nloc, alpha, ld, sem = _grads_matern(dist.group.WORLD)
    q.put((rank, flat, nloc, alpha, ld, sem))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_preconditioned_path():
    """Probe sharding when the solves are conjugate gradients preconditioned by the factorisation on
    the larger basis (Matern rows; csrc/rl_solve.hip hz_*): every rank builds its replica's basis,
    table and map, solves alpha and its share of the probes; the same bits on both ranks, the
    one-rank step's gradient to 1e-7 of its largest entry (solves to 1e-9 -- a system's
    iterates do not depend on which systems share its batch, its exit does not either)."""
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    saved = {k: os.environ.get(k) for k in ('RUNLMC_DEBUG', 'RUNLMC_PRECOND_HI_MIN')}
    try:
        ref, nall, alpha_ref, ld_ref, sem_ref = _grads_matern()
    finally:
        _lib.use_library(None)
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_matern, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][2] + got[1][2] == nall
    assert np.array_equal(got[0][3], got[1][3])            # alpha: rank 0's, broadcast
    assert np.array_equal(got[0][1], got[1][1])            # the assembled gradient
    assert np.abs(got[0][1] - ref).max() < 1e-7 * np.abs(ref).max()
    assert np.abs(got[0][3] - alpha_ref).max() < 1e-8 * np.abs(alpha_ref).max()
    # the preconditioned log det: its 16 probes dealt to the ranks, one all-reduce of (sum, sum of
    # squares) -- the same value on both ranks, the one-rank value to roundoff
    assert got[0][4] == got[1][4] and got[0][5] == got[1][5]
    assert abs(got[0][4] - ld_ref) <= 1e-9 * abs(ld_ref) and abs(got[0][5] - sem_ref) <= 1e-6 * sem_ref


def _grads_n(n_probes):
    """The step on `lmc_small` with a seeded matrix of n_probes probes (every rank draws the
    same matrix and keeps its own rows: rank, rank + world, ...)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import parity_suite as ps
    from cases import Case
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    c = Case('lmc_small')
    fk, K, gk = ps.build_operator(c)
    ad = (0,)
    rs = np.random.RandomState(77).randint(0, 2, (n_probes, c.n)) * 2 - 1
    svc = StochasticDerivService(None, None, n_probes, 1e-4)
    lik = ApproxLMCLikelihood(fk, K, {ad: c.grid_dists}, {ad: (c.W, c.WT)},
                              c.Ys, svc, probes=rs)
    flat = np.concatenate([np.ravel(g) for g in lik.coreg_vec_gradients()] +
                          [np.ravel(g) for g in lik.coreg_diags_gradients()] +
                          [np.ravel(g) for g in lik.kernel_gradients()] +
                          [lik.noise_gradient()])
    return flat, lik.deriv.rs_dev.shape[0], lik.deriv.alpha.copy()


def _worker_n(rank, world, port, q, n_probes):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    flat, nloc, alpha = _grads_n(n_probes)
    q.put((rank, flat, nloc, alpha))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_probes', [5, 128])
def test_eight_rank_probe_sharding(n_probes):
    """The 8-GPU layout of BASELINE config 5 on CPU (gloo, kernels under the emulator;
    reference axis runlmc/lmc/stochastic_deriv.py:39-52): N = 5 probes over 8 ranks -- three
    ranks own NO probe and still solve alpha, take part in the broadcast and the all-reduce
    -- and N = 128, the C5 deal of 16 probes + y per rank.  alpha and the assembled gradient
    are the SAME BITS on all eight ranks and agree with the one-rank step."""
    world = 8
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    try:
        ref, nall, alpha_ref = _grads_n(n_probes)
    finally:
        _lib.use_library(None)
    assert nall == n_probes
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, q, n_probes))
             for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    owned = [g[2] for g in got]
    assert sum(owned) == n_probes                              # every probe owned once
    assert owned == [len(range(r, n_probes, world)) for r in range(world)]
    if n_probes < world:
        assert owned.count(0) == world - n_probes              # ranks with y only
    for g in got[1:]:
        assert np.array_equal(g[3], got[0][3])                 # alpha: rank 0's, broadcast
        assert np.array_equal(g[1], got[0][1])                 # gradient: one all-reduce
    assert np.abs(got[0][3] - alpha_ref).max() < 1e-6 * np.abs(alpha_ref).max()
    scale = np.abs(ref).max()
    assert np.abs(got[0][1] - ref).max() < 1e-4 * scale


def _grads_structured(world_group=None):
    """The same on a synthetic model whose operator runs in the STRUCTURED forms (polynomial
    form verified at rank 24, solver rounds in the row-polynomial form F M F^T: forced onto
    this small system by the batch gate and RUNLMC_STAGED_WT)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from runlmc_amd.util import synth
    from runlmc_amd.lmc.grid_kernel import gen_grid_kernel
    from runlmc_amd.lmc.likelihood import ApproxLMCLikelihood
    from runlmc_amd.lmc.stochastic_deriv import StochasticDerivService
    p = synth.make_problem(3, 2, 1, 700, eps=1.0)
    p.noise = p.noise + 0.5
    fk = synth.functional_kernel(p)
    ad = (0,)
    K, gks = gen_grid_kernel(fk, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.lens)
    assert K.device_operator().grid.top_forms() == ([1, 1], True)
    rs = np.random.RandomState(6).randint(0, 2, (5, p.n)) * 2 - 1
    svc = StochasticDerivService(None, None, len(rs), 1e-6, group=world_group)
    lik = ApproxLMCLikelihood(fk, K, {ad: p.grid_dists}, {ad: (p.W, p.WT)}, p.Ys, svc, probes=rs)
    flat = np.concatenate([np.ravel(g) for g in lik.coreg_vec_gradients()] +
                          [np.ravel(g) for g in lik.coreg_diags_gradients()] +
                          [np.ravel(g) for g in lik.kernel_gradients()] +
                          [lik.noise_gradient()])
    return flat, lik.deriv.rs_dev.shape[0], lik.deriv.alpha.copy()


_STRUCTURED_ENV = {'RUNLMC_STAGED_WT': '1', 'RUNLMC_NO_FUSE_W': '1', 'RUNLMC_NO_FUSE_WT': '1',
                   'RUNLMC_LR_MIN': '0', 'RUNLMC_NO_POLY_ROUND': '1'}


def _worker_structured(rank, world, port, q):
    os.environ.update(_STRUCTURED_ENV)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    flat, nloc, alpha = _grads_structured()
    q.put((rank, flat, nloc, alpha))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_probe_sharding_structured_forms(monkeypatch):
    """Two ranks with 3 and 2 of 5 probes (batches of 5 and 3 right-hand sides, both through
    the small-batch projection of the row-polynomial rounds): alpha and the assembled
    gradient are the SAME BITS on both ranks (rank 0's alpha broadcast, one all-reduce) and
    agree with the one-rank step to the solver's tolerance."""
    for k, v in _STRUCTURED_ENV.items():
        monkeypatch.setenv(k, v)
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    try:
        ref, nall, alpha_ref = _grads_structured()
    finally:
        _lib.use_library(None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_structured, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][2] + got[1][2] == nall and got[0][2] == 3
    assert np.array_equal(got[0][3], got[1][3])
    assert np.array_equal(got[0][1], got[1][1])
    assert np.abs(got[0][3] - alpha_ref).max() < 1e-6 * np.abs(alpha_ref).max()
    assert np.abs(got[0][1] - ref).max() < 1e-4 * np.abs(ref).max()


def _block_solve():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import parity_suite as ps
    from cases import Case
    from runlmc_amd.approx.iterative import Iterative
    c = Case('lmc_c1')
    fk, K, gk = ps.build_operator(c)
    rng = np.random.RandomState(9)
    B = np.vstack([c.y] + [rng.randn(c.n) for _ in range(4)])      # 5 right-hand sides
    return Iterative.solve_sharded(K, B, minres=False, tol=1e-4)    # CG, as config 4


def _worker_block(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from runlmc_amd import _lib, build
    _lib.use_library(build.EMU_LIB)
    X, it, rs = _block_solve()
    q.put((rank, X, it, rs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_of_right_hand_sides():
    """BASELINE config 4's axis: a block of right-hand sides (CG) split over
    ranks, gathered with one all-reduce; both ranks end with every solution,
    equal to the one-process block solve."""
    from runlmc_amd import _lib, build
    _lib.use_library(build.build_emu())
    try:
        Xr, itr, rsr = _block_solve()
    finally:
        _lib.use_library(None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_block, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    assert np.all(got[0][3] < 1e-4) and np.all(rsr < 1e-4)
    # (which vectors share a transform differs between the two partitions)
    assert np.abs(got[0][1] - Xr).max() < 1e-6 * np.abs(Xr).max()
    assert np.all(np.abs(got[0][2] - itr) <= np.maximum(6, itr // 5))


def test_shard_rows_partition():
    from runlmc_amd.util.dist import shard_rows, rank_world
    assert rank_world() == (0, 1)
    assert shard_rows(5) == [0, 1, 2, 3, 4]
