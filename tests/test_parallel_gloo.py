"""world_size-2 gloo test of the walker-sharding path (CPU; the rank-local
evaluator is a stand-in -- the oracle -- because there is no GPU here and the
product has no CPU path)."""
import os
import socket
import subprocess
import sys

import numpy as np

import helpers
from psfmc_amd.parallel import shard_bounds

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path[:0] = [%(root)r, %(root)r + '/oracle', %(root)r + '/tools', %(root)r + '/tests']
import helpers
from psfmc_amd.parallel import ShardedLogPosterior
dist.init_process_group('gloo')
case = helpers.load_case('synth128x2')
field = helpers.oracle_field(case)
layout = helpers.LAYOUT['synth128x2']
calls = []
def evaluate(theta):
    calls.append(len(theta))
    return np.array([helpers.oracle_loglike(field, layout, t) for t in theta])
theta = case['params'][:7]
out = ShardedLogPosterior(evaluate)(theta)
np.save(os.path.join(%(out)r, 'rank%%d.npy' %% dist.get_rank()), out)
np.save(os.path.join(%(out)r, 'calls%%d.npy' %% dist.get_rank()), np.array(calls))
dist.destroy_process_group()
'''


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_all_gather(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': root, 'out': str(tmp_path)})
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    subprocess.check_call(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        env=env, timeout=300)
    case = helpers.load_case('synth128x2')
    r0 = np.load(tmp_path / 'rank0.npy')
    r1 = np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1)                      # every rank has the full vector
    assert helpers.rel_err(r0, case['loglike_f64'][:7]) <= 1e-12
    assert np.load(tmp_path / 'calls0.npy').tolist() == [4]
    assert np.load(tmp_path / 'calls1.npy').tolist() == [3]
