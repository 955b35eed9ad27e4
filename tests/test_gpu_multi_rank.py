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
          '--no-extra']


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
        g1, g2 = np.array(a['grad_sample']), np.array(b['grad_sample'])
        assert np.abs(g1 - g2).max() <= 1e-4 * np.abs(g1).max(), (g1, g2)
        assert abs(a['grad_norm'] - b['grad_norm']) <= 1e-4 * a['grad_norm']


def test_bench_refuses_wrong_world_size():
    """--gpus N without N ranks must fail loudly, with the launch line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'] + COMMON,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert 'torch.distributed.run' in (r.stderr + r.stdout)
