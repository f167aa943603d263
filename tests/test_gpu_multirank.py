"""Multi-rank GPU tests on the ONE-GPU box: two ranks (gloo rendezvous, both on device 0)
drive the REAL engine -- the sharded log-posterior, the device-resident sampler with every
half-step's proposals sharded over the ranks, and `model_galaxy_mcmc` end to end -- and must
reproduce the single-rank results bit for bit (per-walker log-posteriors do not depend on
the batch they are evaluated in).  The RCCL (`nccl`) data path differs only in where the
gathered tensor lives (parallel.RankGroup.host_staged); the driver exercises it at N > 1."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
import synth_field

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ITER = 5

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
root, out = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, root + '/oracle', root + '/tools', root + '/tests']
import helpers, synth_field
from psfmc_amd.parallel import ShardedLogPosterior, RankGroup
from psfmc_amd.sampler import DeviceEnsembleSampler
from psfmc_amd import model_galaxy_mcmc
dist.init_process_group('gloo')
rank = dist.get_rank()
torch.cuda.set_device(0)
case = helpers.load_case('synth256')
work = os.path.join(out, 'w%d' % rank)
os.makedirs(work)
model = helpers.build_model('synth256', case, work, backend='fused', max_walkers=128)
# (a) sharded evaluation of the golden vectors (65 walkers: uneven blocks)
sharded = ShardedLogPosterior(model)
lnp = sharded(case['params'])
np.save(os.path.join(out, 'lnp%d.npy' % rank), lnp)
# (a') the same with the vectors resident on the GPU (what `bench.py --gpus N` times): device tensor in, the
# all-gathered device tensor out; a block larger than the context's max_walkers goes through in slices
theta_dev = torch.from_numpy(np.ascontiguousarray(case['params'])).to('cuda:0')
lnp_dev = sharded.evaluate_device(theta_dev)
big = torch.from_numpy(np.ascontiguousarray(np.tile(case['params'], (5, 1)))).to('cuda:0')     # 325 walkers: blocks > 128
lnp_big = sharded.evaluate_device(big)
torch.cuda.synchronize()
np.save(os.path.join(out, 'lnpdev%d.npy' % rank), lnp_dev.cpu().numpy())
np.save(os.path.join(out, 'lnpbig%d.npy' % rank), lnp_big.cpu().numpy())
# (b) the device-resident sampler, half-steps sharded, images accumulated per rank
p0 = synth_field.draw_walkers(256, 1, __NWALK__, seed=77, near_truth=case['params'][-1])
samp = DeviceEnsembleSampler(__NWALK__, model, group=True, block=3, accumulate=True)
samp.random_state = np.random.RandomState(123).get_state()
for res in samp.sample(p0, iterations=__NITER__):
    pass
np.save(os.path.join(out, 'chain%d.npy' % rank), samp.chain)
np.save(os.path.join(out, 'lnchain%d.npy' % rank), samp.lnprobability)
np.save(os.path.join(out, 'nacc%d.npy' % rank), samp.naccepted)
own = model.accumulated_samples
model.reduce_accumulated(samp.ranks)
post = model.collect_posterior_images()
np.savez(os.path.join(out, 'post%d.npz' % rank), count=model.accumulated_samples, own=own, **post)
model.close()
# (c) the entry point end to end on the reference's example field
np.random.seed(3 + rank)                 # rank 0's start positions are broadcast
mfile = os.path.join(root, 'tests', 'golden', 'example', 'model_example.py')
shared = os.path.join(out, 'mcmc')
os.makedirs(shared, exist_ok=True)
m, db = model_galaxy_mcmc(mfile, output_name=os.path.join(shared, 'run'), iterations=6, burn=3, chains=40,
                          random_state=5, quiet=True, write_fits=('convolved_model',))
np.save(os.path.join(out, 'dbln%d.npy' % rank), np.asarray(db['lnprobability']))
m.close()
dist.destroy_process_group()
'''


@pytest.mark.parametrize('world,N_WALK', [(2, 48), (3, 40)])      # 40 walkers on 3 ranks: blocks of 7, 7, 6
def test_ranks_reproduce_one_rank(tmp_path, world, N_WALK):
    from psfmc_amd import model_galaxy_mcmc, fits_io
    from psfmc_amd.database import load_database
    from psfmc_amd.sampler import DeviceEnsembleSampler
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.replace('__NWALK__', str(N_WALK)).replace('__NITER__', str(N_ITER)))
    out = tmp_path / 'out'
    out.mkdir()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    subprocess.check_call(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), ROOT, str(out)],
        env=env, timeout=600)

    case = helpers.load_case('synth256')
    # (a) gathered log-posteriors: identical on both ranks, equal to the reference's
    l0 = np.load(out / 'lnp0.npy')
    for r in range(1, world):
        assert np.array_equal(l0, np.load(out / ('lnp%d.npy' % r)))
    assert helpers.rel_err(l0, case['lnprob']) <= 1e-6
    single = tmp_path / 'single'
    single.mkdir()
    model = helpers.build_model('synth256', case, single, backend='fused', max_walkers=128)
    assert np.array_equal(model.log_posterior_batch(case['params']), l0)      # bitwise: batch independent
    for r in range(world):
        assert np.array_equal(np.load(out / ('lnpdev%d.npy' % r)), l0), r          # (a') device-resident route
        assert np.array_equal(np.load(out / ('lnpbig%d.npy' % r)), np.tile(l0, 5)), r
    # (b) the two-rank chain is the one-rank chain
    p0 = synth_field.draw_walkers(256, 1, N_WALK, seed=77, near_truth=case['params'][-1])
    samp = DeviceEnsembleSampler(N_WALK, model, block=3, accumulate=True)
    samp.random_state = np.random.RandomState(123).get_state()
    for _ in samp.sample(p0, iterations=N_ITER):
        pass
    for r in range(world):
        assert np.array_equal(np.load(out / ('chain%d.npy' % r)), samp.chain), r
        assert np.array_equal(np.load(out / ('lnchain%d.npy' % r)), samp.lnprobability), r
        assert np.array_equal(np.load(out / ('nacc%d.npy' % r)), samp.naccepted), r
    assert samp.naccepted.sum() > 0
    post = model.collect_posterior_images()
    from psfmc_amd.parallel import shard_bounds
    for r in range(world):
        got = np.load(out / ('post%d.npz' % r))
        assert int(got['count']) == N_WALK * N_ITER == model.accumulated_samples
        lo, hi = shard_bounds(N_WALK, world, r)
        assert int(got['own']) == (hi - lo) * N_ITER            # each rank summed its block of the walkers
        for kind, img in post.items():
            scale = np.abs(img[np.isfinite(img)]).max()
            assert np.abs(got[kind] - img).max() <= 1e-12 * scale, (r, kind)
    model.close()
    # (c) the entry point: same database from two ranks and from one process
    np.random.seed(3)
    mfile = os.path.join(ROOT, 'tests', 'golden', 'example', 'model_example.py')
    m, db = model_galaxy_mcmc(mfile, output_name=str(single / 'run'), iterations=6, burn=3, chains=40,
                              random_state=5, quiet=True, write_fits=('convolved_model',), group=None)
    m.close()
    ref = np.asarray(db['lnprobability'])
    for r in range(world):
        assert np.array_equal(np.load(out / ('dbln%d.npy' % r)), ref), r
    two = load_database(str(out / 'mcmc' / 'run_db.fits'))
    for name in db.colnames:
        assert np.array_equal(two[name], db[name]), name
    a = fits_io.read_image(str(out / 'mcmc' / 'run_convolved_model.fits'))
    b = fits_io.read_image(str(single / 'run_convolved_model.fits'))
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


RCCL_WORKER = r'''
import json, os, sys, time
import numpy as np
import torch, torch.distributed as dist
root, out = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, root + '/oracle', root + '/tools', root + '/tests']
import helpers, synth_field
from psfmc_amd.parallel import ShardedLogPosterior, RankGroup
from psfmc_amd.sampler import DeviceEnsembleSampler
from psfmc_amd import model_galaxy_mcmc
# the RCCL back end with a world of ONE rank; deliberately NO torch.cuda.set_device here: RankGroup
# itself must put the collectives on its device (round-2 advice)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
rg = RankGroup(None, 'cuda:0', shortcut=False)
assert rg.active and rg.world == 1 and not rg.single and not rg.host_staged and rg.backend == 'nccl'
case = helpers.load_case('synth256')
work = os.path.join(out, 'w')
os.makedirs(work)
model = helpers.build_model('synth256', case, work, backend='fused', max_walkers=128)
# (a) ShardedLogPosterior: device tensors through all_gather_into_tensor + index_select
sharded = ShardedLogPosterior(model, group=rg)
lnp = sharded(case['params'])
np.save(os.path.join(out, 'lnp.npy'), lnp)
t0 = time.perf_counter()
for _ in range(20):
    sharded(case['params'])
per_call = (time.perf_counter() - t0) / 20
# (b) object broadcast and the host-array all-reduce of the posterior sums
assert rg.broadcast_object({'a': 1, 'b': [2.5]}) == {'a': 1, 'b': [2.5]}
assert np.array_equal(rg.all_reduce_sum_host(np.arange(5.0)), np.arange(5.0))
# (c) the device-resident sampler with every half-step's gather going through RCCL
p0 = synth_field.draw_walkers(256, 1, __NWALK__, seed=77, near_truth=case['params'][-1])
samp = DeviceEnsembleSampler(__NWALK__, model, group=rg, block=3, accumulate=True)
assert samp.ranks is rg
samp.random_state = np.random.RandomState(123).get_state()
t0 = time.perf_counter()
for res in samp.sample(p0, iterations=__NITER__):
    pass
per_iter = (time.perf_counter() - t0) / __NITER__
np.save(os.path.join(out, 'chain.npy'), samp.chain)
np.save(os.path.join(out, 'lnchain.npy'), samp.lnprobability)
model.reduce_accumulated(rg)
post = model.collect_posterior_images()
np.savez(os.path.join(out, 'post.npz'), count=model.accumulated_samples, **post)
model.close()
# (d) the entry point with the group handed in
np.random.seed(3)
mfile = os.path.join(root, 'tests', 'golden', 'example', 'model_example.py')
m, db = model_galaxy_mcmc(mfile, output_name=os.path.join(out, 'run'), iterations=6, burn=3, chains=40,
                          random_state=5, quiet=True, write_fits=('convolved_model',), group=rg)
np.save(os.path.join(out, 'dbln.npy'), np.asarray(db['lnprobability']))
m.close()
with open(os.path.join(out, 'timing.json'), 'w') as f:
    json.dump({'sharded_call_ms_65_walkers': per_call * 1e3, 'sampler_iteration_ms': per_iter * 1e3}, f)
dist.destroy_process_group()
'''


def test_rccl_path_with_one_rank(tmp_path):
    """The RCCL (`nccl`) branch of the multi-GPU path, exercised on the one GPU there is: a process
    group of ONE rank with the world-of-one short-cuts switched off (`RankGroup(shortcut=False)`),
    so that `all_gather_into_tensor` on device tensors, `index_select`, `broadcast_object_list`,
    `all_reduce` and the stream hand-over really execute through RCCL -- chain, database and
    posterior images must equal the plain single-process ones bit for bit."""
    from psfmc_amd import model_galaxy_mcmc
    from psfmc_amd.sampler import DeviceEnsembleSampler
    n_walk = 48
    script = tmp_path / 'worker.py'
    script.write_text(RCCL_WORKER.replace('__NWALK__', str(n_walk)).replace('__NITER__', str(N_ITER)))
    out = tmp_path / 'out'
    out.mkdir()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port))
    subprocess.check_call([sys.executable, str(script), ROOT, str(out)], env=env, timeout=600)

    case = helpers.load_case('synth256')
    single = tmp_path / 'single'
    single.mkdir()
    model = helpers.build_model('synth256', case, single, backend='fused', max_walkers=128)
    assert np.array_equal(model.log_posterior_batch(case['params']), np.load(out / 'lnp.npy'))
    p0 = synth_field.draw_walkers(256, 1, n_walk, seed=77, near_truth=case['params'][-1])
    samp = DeviceEnsembleSampler(n_walk, model, block=3, accumulate=True)
    samp.random_state = np.random.RandomState(123).get_state()
    for _ in samp.sample(p0, iterations=N_ITER):
        pass
    assert np.array_equal(np.load(out / 'chain.npy'), samp.chain)
    assert np.array_equal(np.load(out / 'lnchain.npy'), samp.lnprobability)
    post = model.collect_posterior_images()
    got = np.load(out / 'post.npz')
    assert int(got['count']) == n_walk * N_ITER
    for kind, img in post.items():
        assert np.abs(got[kind] - img).max() <= 1e-12 * np.abs(img[np.isfinite(img)]).max(), kind
    model.close()
    np.random.seed(3)
    mfile = os.path.join(ROOT, 'tests', 'golden', 'example', 'model_example.py')
    m, db = model_galaxy_mcmc(mfile, output_name=str(single / 'run'), iterations=6, burn=3, chains=40,
                              random_state=5, quiet=True, write_fits=('convolved_model',), group=None)
    m.close()
    assert np.array_equal(np.load(out / 'dbln.npy'), np.asarray(db['lnprobability']))
    print(open(out / 'timing.json').read())


@pytest.mark.skipif(__import__('torch').cuda.device_count() < 2, reason='needs two GPUs')
def test_model_galaxy_mcmc_two_gpus_rccl(tmp_path):
    """`model_galaxy_mcmc` over RCCL on two real GPUs (skipped on the one-GPU box): rank r passes
    device=r and does not call torch.cuda.set_device -- the entry point has to place its collectives."""
    worker = tmp_path / 'w.py'
    worker.write_text(r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
root, out = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, root + '/oracle', root + '/tools', root + '/tests']
from psfmc_amd import model_galaxy_mcmc
local = int(os.environ['LOCAL_RANK'])
dist.init_process_group('nccl', device_id=torch.device('cuda', local))
np.random.seed(3 + local)
mfile = os.path.join(root, 'tests', 'golden', 'example', 'model_example.py')
m, db = model_galaxy_mcmc(mfile, output_name=os.path.join(out, 'run'), iterations=6, burn=3, chains=40,
                          random_state=5, quiet=True, write_fits=('convolved_model',), device=local)
np.save(os.path.join(out, 'dbln%d.npy' % local), np.asarray(db['lnprobability']))
m.close()
dist.destroy_process_group()
""")
    out = tmp_path / 'out'
    out.mkdir()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    subprocess.check_call(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(worker), ROOT, str(out)],
        env=env, timeout=600)
    from psfmc_amd import model_galaxy_mcmc
    np.random.seed(3)
    mfile = os.path.join(ROOT, 'tests', 'golden', 'example', 'model_example.py')
    m, db = model_galaxy_mcmc(mfile, output_name=str(tmp_path / 'one'), iterations=6, burn=3, chains=40,
                              random_state=5, quiet=True, write_fits=('convolved_model',), group=None)
    m.close()
    for r in range(2):
        assert np.array_equal(np.load(out / ('dbln%d.npy' % r)), np.asarray(db['lnprobability'])), r
