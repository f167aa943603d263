#!/usr/bin/env python3
"""Wall time per MCMC iteration of the host-loop and device-resident samplers
(256^2 field, 1 PS + 1 Sersic).  Run on the GPU box."""
import os
import sys
import tempfile
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers  # noqa: E402
from psfmc_amd.sampler import EnsembleSampler, DeviceEnsembleSampler  # noqa: E402

case = helpers.load_case('synth256')
for n_w in (64, 256, 1024):
    m = helpers.build_model('synth256', case, tempfile.mkdtemp(), max_walkers=n_w)
    np.random.seed(1)
    p0 = m.init_params_from_priors(n_w)
    for name, s in (('host loop', EnsembleSampler(n_w, m.num_params, batch_lnpostfn=m.log_posterior_batch)),
                    ('device', DeviceEnsembleSampler(n_w, m, block=100))):
        s.random_state = np.random.RandomState(5).get_state()
        list(s.sample(p0, iterations=100))          # warm-up (first large RNG draw is slow)
        t = time.perf_counter()
        list(s.sample(p0, iterations=100))
        dt = time.perf_counter() - t
        print('%4d walkers  %-9s  %.3f ms/iteration  %8.0f evals/s' % (n_w, name, dt * 10, 100 * n_w / dt))
    m.close()
