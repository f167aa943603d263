"""Posterior-image accumulation on a transform of more than 256 pixels per row (the rasteriser's power-table
form in k_raster_sums): ms per MCMC iteration of the device sampler with and without the per-iteration image sums.
usage: python3 tools/time_accumulation_large.py [side] [sersic]   (PSFMC_LIB selects the library build)"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
from test_gpu_fullsize import make_model
from psfmc_amd.sampler import DeviceEnsembleSampler
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_sersic = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for n_w in (64, 256):
    m, fld = make_model(side, n_sersic, 'fused', max_walkers=n_w)
    np.random.seed(1)
    p0 = m.init_params_from_priors(n_w)
    for acc in (False, True):
        s = DeviceEnsembleSampler(n_w, m, block=20, accumulate=acc)
        s.random_state = np.random.RandomState(5).get_state()
        list(s.sample(p0, iterations=20))
        t = time.perf_counter()
        list(s.sample(p0, iterations=60))
        dt = time.perf_counter() - t
        print('%s %d^2 %d Sersic %4d walkers accumulate=%-5s %.3f ms/iteration' % (
            os.path.basename(os.environ.get('PSFMC_LIB', 'in-tree')), side, n_sersic, n_w, acc, dt * 1e3 / 60), flush=True)
    m.close()
