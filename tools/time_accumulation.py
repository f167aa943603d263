import os, sys, tempfile, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers
from psfmc_amd.sampler import DeviceEnsembleSampler
for n_w in (22, 64, 256):
    case = helpers.load_case('synth256')
    m = helpers.build_model('synth256', case, tempfile.mkdtemp(), max_walkers=n_w)
    np.random.seed(1)
    p0 = m.init_params_from_priors(n_w)
    for acc in (False, True):
        s = DeviceEnsembleSampler(n_w, m, block=50, accumulate=acc)
        s.random_state = np.random.RandomState(5).get_state()
        list(s.sample(p0, iterations=50))
        t = time.perf_counter()
        list(s.sample(p0, iterations=200))
        dt = time.perf_counter() - t
        print('%4d walkers accumulate=%-5s %.3f ms/iteration' % (n_w, acc, dt * 1e3 / 200))
    m.close()
