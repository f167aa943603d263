#!/usr/bin/env python3
"""Device-resident sampler only, one walker count (argv[1], default 256): the target of
`rocprofv3 --kernel-trace --stats -- python3 tests/profile_sampler.py 256`."""
import os
import sys
import tempfile
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers  # noqa: E402
from psfmc_amd.sampler import DeviceEnsembleSampler  # noqa: E402

n_w = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
case = helpers.load_case('synth256')
m = helpers.build_model('synth256', case, tempfile.mkdtemp(), max_walkers=n_w)
np.random.seed(1)
p0 = m.init_params_from_priors(n_w)
if os.environ.get('PSFMC_CHUNK'):
    m.engine.set_option('chunk_walkers', int(os.environ['PSFMC_CHUNK']))
if os.environ.get('PSFMC_GRAPH'):
    m.engine.set_option('graph', int(os.environ['PSFMC_GRAPH']))
s = DeviceEnsembleSampler(n_w, m, block=100)
s.random_state = np.random.RandomState(5).get_state()
list(s.sample(p0, iterations=100))
t = time.perf_counter()
list(s.sample(p0, iterations=iters))
dt = time.perf_counter() - t
print('%4d walkers  device  %.3f ms/iteration  %8.0f evals/s' % (n_w, dt * 1e3 / iters, iters * n_w / dt))
m.close()
