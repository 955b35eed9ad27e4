"""The benchmark's multi-rank path end to end on ONE GPU: two ranks launched by
torch.distributed.run share cuda:0 and talk over gloo (the driver's 8-GPU run
uses the same code with backend nccl = RCCL, one GPU per rank).  Checks the
JSON contract (n_gpus, weak-scaling value, strong-scaling NLL+gradient) and
that the probe-sharded gradient equals the one-rank gradient (same probes,
different batching -> solver-level noise, 1e-4).  Partition semantics:
reference runlmc/lmc/stochastic_deriv.py:39-52 (N+1 independent solves)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--config', 'c2', '--steps', '20', '--warmup', '3', '--no-cpu', '--no-sweep',
          '--no-extra', '--no-families']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, OMP_NUM_THREADS='1')
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + COMMON,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                         timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = _last_json(one.stdout)
    two = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
         os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--same-gpu']
        + COMMON, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    j2 = _last_json(two.stdout)
    assert j2['n_gpus'] == 2 and j1['n_gpus'] == 1
    assert j2['scaling'] == 'weak' and j2['config']['parallelism'] == 'probe-shard x2'
    assert j2['value'] > 0 and j2['steps'] == 20 and j2['warmup'] == 3
    for key in ('nll_grad', 'nll_grad_eps1'):
        a, b = j1[key], j2[key]
        assert b['scaling'] == 'strong' and b['probes_per_rank'] == 8
        assert b['n_probes_global'] == a['n_probes_global'] == 16
        # alpha and the assembled gradient: the same BITS on both ranks (checked inside
        # the run over the ranks' checksums); against the one-rank run the batching
        # differs, hence solver-level noise
        assert b['bits_equal_across_ranks'] == {'alpha': True, 'gradient': True}
        g1, g2 = np.array(a['grad_sample']), np.array(b['grad_sample'])
        assert np.abs(g1 - g2).max() <= 1e-4 * np.abs(g1).max(), (g1, g2)
        assert abs(a['grad_norm'] - b['grad_norm']) <= 1e-4 * a['grad_norm']


def test_collectives_through_rccl_in_a_world_of_one():
    """One rank, process group on backend nccl (= RCCL), collectives forced: the
    step's alpha broadcast and gradient all-reduce run on RCCL itself on a
    one-GPU box, and the step's numbers equal the plain one-rank run's bit for
    bit (a world of one changes no arithmetic)."""
    env = dict(os.environ, OMP_NUM_THREADS='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()))
    args = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + COMMON
    plain = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                           timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    forced = subprocess.run(args + ['--force-dist', '--dist-backend', 'nccl'],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                            timeout=900)
    assert forced.returncode == 0, forced.stderr[-3000:]
    j0, j1 = _last_json(plain.stdout), _last_json(forced.stdout)
    assert j1['collectives']['backend'] == 'nccl' and j1['collectives']['forced_in_world_of_one']
    for key in ('nll_grad', 'nll_grad_eps1'):
        assert j0[key]['grad_sample'] == j1[key]['grad_sample'], (j0[key], j1[key])
        assert j0[key]['grad_norm'] == j1[key]['grad_norm']


def test_sharded_block_through_rccl_in_a_world_of_one():
    """Iterative.solve_sharded's all-gather on RCCL with one rank."""
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from runlmc_amd.util import dist as rdist, synth\n"
        "rdist.force_collectives(True)\n"
        "from runlmc_amd.lmc.grid_kernel import gen_grid_kernel\n"
        "from runlmc_amd.approx.iterative import Iterative\n"
        "p = synth.make_problem(3, 2, 1, 400, eps=1.0)\n"
        "fk = synth.functional_kernel(p)\n"
        "K, _ = gen_grid_kernel(fk, {(0,): p.grid_dists}, {(0,): (p.W, p.WT)}, p.lens)\n"
        "B = np.random.RandomState(0).randn(5, p.n)\n"
        "a = Iterative.solve_sharded(K, B, tol=1e-4, group=dist.group.WORLD)[0]\n"
        "b = Iterative.solve(K, B, tol=1e-4)\n"
        "assert np.array_equal(np.asarray(a), np.asarray(b))\n"
        "dist.destroy_process_group()\n"
        "print('ok')\n") % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-3000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher: the script starts its two ranks itself
    (a torch.distributed.run child; here both on cuda:0 over gloo), relays rank 0's line and
    exits with the children's status; the line relates the sharded step to the one-rank step."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--dist-backend', 'gloo', '--same-gpu'] + COMMON,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                       timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j['n_gpus'] == 2 and j['config']['parallelism'] == 'probe-shard x2'
    assert j['nll_grad']['speedup_vs_one_rank'] > 0 and j['nll_grad']['one_rank_seconds'] > 0
    assert j['nll_grad']['bits_equal_across_ranks'] == {'alpha': True, 'gradient': True}


def test_bench_refuses_a_launcher_of_another_size():
    """Under a launcher whose WORLD_SIZE is not --gpus the script fails loudly, with the
    launch line (it does not start ranks inside ranks)."""
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'] + COMMON,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=env)
    assert r.returncode != 0
    assert 'torch.distributed.run' in (r.stderr + r.stdout)
