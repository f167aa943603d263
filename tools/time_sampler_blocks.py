import os, sys, tempfile, time
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
import helpers
from psfmc_amd.sampler import DeviceEnsembleSampler
case = helpers.load_case('synth256')
for n_w in (64, 256):
    m = helpers.build_model('synth256', case, tempfile.mkdtemp(), max_walkers=n_w)
    np.random.seed(1)
    p0 = m.init_params_from_priors(n_w)
    s = DeviceEnsembleSampler(n_w, m, block=64)
    s.random_state = np.random.RandomState(5).get_state()
    lnp = m.log_posterior_batch(p0)
    nacc = np.zeros(n_w, dtype=np.int64)
    t = time.perf_counter(); draws, states = s._draw(64); t_draw = time.perf_counter() - t
    eng = m.engine
    eng.stretch_run(p0, lnp, *draws, nacc, store=True)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); eng.stretch_run(p0, lnp, *draws, nacc, store=True); ts.append(time.perf_counter() - t)
    t = time.perf_counter(); out = list(s.sample(p0, lnprob0=lnp, iterations=640)); t_all = time.perf_counter() - t
    print('%4d walkers: draw(64) %.2f ms, stretch_run(64 iterations) %.2f ms = %.3f ms/iter, sample(640) %.3f ms/iter'
          % (n_w, t_draw * 1e3, min(ts) * 1e3, min(ts) * 1e3 / 64, t_all * 1e3 / 640))
    m.close()

# where the host time of sample() goes (256 walkers)
import cProfile, pstats
m = helpers.build_model('synth256', case, tempfile.mkdtemp(), max_walkers=256)
np.random.seed(1)
p0 = m.init_params_from_priors(256)
s = DeviceEnsembleSampler(256, m, block=64)
lnp = m.log_posterior_batch(p0)
list(s.sample(p0, lnprob0=lnp, iterations=64))
pr = cProfile.Profile(); pr.enable()
list(s.sample(p0, lnprob0=lnp, iterations=640))
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
m.close()
