import os, sys, shutil, tempfile, time, cProfile, pstats
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers
from psfmc_amd import model_galaxy_mcmc
src = os.path.join(helpers.GOLDEN, 'example')
tmp = tempfile.mkdtemp()
for name in os.listdir(src):
    if os.path.isfile(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), tmp)
mf = os.path.join(tmp, 'model_example.py')
np.random.seed(5)
model_galaxy_mcmc(mf, output_name=os.path.join(tmp, 'warm'), iterations=20, burn=20, chains=256, random_state=11, quiet=True)
pr = cProfile.Profile()
pr.enable()
model_galaxy_mcmc(mf, output_name=os.path.join(tmp, 'soak'), iterations=3000, burn=1000, chains=256, random_state=11, quiet=True)
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
