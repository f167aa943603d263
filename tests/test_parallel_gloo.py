"""gloo tests (world sizes 2 and 8 -- the node BASELINE configs 4 and 5 name) of the walker-sharding
path (CPU; the rank-local evaluator is a stand-in -- the oracle -- because there is no GPU here and
the product has no CPU path)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
from psfmc_amd.parallel import shard_bounds

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path[:0] = [%(root)r, %(root)r + '/oracle', %(root)r + '/tools', %(root)r + '/tests']
import helpers
import torch
from psfmc_amd.parallel import ShardedLogPosterior, RankGroup
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
# the rank-group primitives the sharded sampler uses: uneven blocks, sum, broadcast
rg = RankGroup()
n = 11
lo, hi = rg.block(n)
send = torch.zeros(rg.slot(n), dtype=torch.float64)
send[:hi - lo] = torch.arange(lo, hi, dtype=torch.float64) * 1.5
full = rg.all_gather_blocks(send, n).numpy()
tot = rg.all_reduce_sum_host(np.full(5, rg.rank + 1.0))
obj = rg.broadcast_object({'rank': rg.rank, 'state': np.arange(3) + rg.rank})
np.savez(os.path.join(%(out)r, 'prim%%d.npz' %% rg.rank), full=full, tot=tot, obj_rank=obj['rank'],
         obj_state=obj['state'], world=rg.world, staged=rg.host_staged)
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


@pytest.mark.parametrize('world', [2, 8])
def test_rank_group_all_gather(tmp_path, world):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': root, 'out': str(tmp_path)})
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    subprocess.check_call(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        env=env, timeout=600)
    case = helpers.load_case('synth128x2')
    ranks = [np.load(tmp_path / ('rank%d.npy' % r)) for r in range(world)]
    for r in ranks[1:]:
        assert np.array_equal(ranks[0], r)             # every rank has the full vector
    assert helpers.rel_err(ranks[0], case['loglike_f64'][:7]) <= 1e-12
    # 7 walkers: contiguous blocks that differ by at most one, ranks past the walkers evaluate nothing
    want = [shard_bounds(7, world, r) for r in range(world)]
    for r, (lo, hi) in enumerate(want):
        assert np.load(tmp_path / ('calls%d.npy' % r)).tolist() == ([hi - lo] if hi > lo else [])
    for r in range(world):
        prim = np.load(tmp_path / ('prim%d.npz' % r))
        assert int(prim['world']) == world and bool(prim['staged'])
        assert np.array_equal(prim['full'], np.arange(11) * 1.5)
        assert np.array_equal(prim['tot'], np.full(5, world * (world + 1) / 2.0))
        assert int(prim['obj_rank']) == 0 and np.array_equal(prim['obj_state'], np.arange(3))
