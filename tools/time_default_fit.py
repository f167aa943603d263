"""The reference's call as a user makes it: model_galaxy_mcmc(model_file) on the example model with the DEFAULT
number of chains (2 P + 2 = 38, psfMC/fitting.py:52-53) -- wall time with and without the device sampler's
whole-iteration launches (library option `speculate`)."""
import os, sys, shutil, tempfile, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers
from psfmc_amd import model_galaxy_mcmc, MultiComponentModel
src = os.path.join(helpers.GOLDEN, 'example')
tmp = tempfile.mkdtemp()
for name in os.listdir(src):
    if os.path.isfile(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), tmp)
mf = os.path.join(tmp, 'model_example.py')
its, burn = 4000, 1000
for label, spec in (('whole-iteration launches (default)', -1), ('two half-steps per iteration', 0)):
    model = MultiComponentModel(mf, max_walkers=1024)
    model.engine.set_option('speculate', spec)
    np.random.seed(5)
    model_galaxy_mcmc(model, output_name=os.path.join(tmp, 'warm%d' % spec), iterations=20, burn=20, random_state=11, quiet=True)
    t0 = time.perf_counter()
    model_galaxy_mcmc(model, output_name=os.path.join(tmp, 'run%d' % spec), iterations=its, burn=burn, random_state=11, quiet=True)
    dt = time.perf_counter() - t0
    n_w = 2 * model.num_params + 2
    print('%-40s %d chains x (%d + %d) iterations = %.0f k evaluations: %.2f s wall (%.3f ms per iteration, database / statistics / images included)'
          % (label, n_w, burn, its, n_w * (its + burn) / 1e3, dt, dt * 1e3 / (its + burn)), flush=True)
    model.close()
