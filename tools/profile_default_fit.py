import os, sys, shutil, tempfile, time, cProfile, pstats
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers
from psfmc_amd import model_galaxy_mcmc, MultiComponentModel
src = os.path.join(helpers.GOLDEN, 'example')
tmp = tempfile.mkdtemp()
for name in os.listdir(src):
    if os.path.isfile(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), tmp)
mf = os.path.join(tmp, 'model_example.py')
model = MultiComponentModel(mf, max_walkers=1024)
np.random.seed(5)
model_galaxy_mcmc(model, output_name=os.path.join(tmp, 'warm'), iterations=20, burn=20, random_state=11, quiet=True)
pr = cProfile.Profile(); pr.enable()
model_galaxy_mcmc(model, output_name=os.path.join(tmp, 'run'), iterations=4000, burn=1000, random_state=11, quiet=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
