"""Device-resident sampler at the reference's default ensemble sizes (chains = 2 P + 2, fitting.py:52-53) and
larger: ms per iteration of psfmc_stretch_run with whole-iteration launches (option `speculate`: one pipeline
pass per iteration over 3 half-ensembles of proposals) forced on and off, for the library's auto rule.
usage: python3 tools/time_small_sampler.py [side:walkers,walkers,... ...]"""
import os, sys, tempfile, time
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [R, R + '/tests', R + '/oracle', R + '/tools']
from test_gpu_fullsize import make_model
from psfmc_amd.sampler import DeviceEnsembleSampler
specs = sys.argv[1:] or ['128:22,40,64,96,128,192,256', '256:22,40,64,96,128', '512:22,40,64']
for spec in specs:
    side, ws = spec.split(':')
    side = int(side)
    for n_w in (int(w) for w in ws.split(',')):
        m, fld = make_model(side, 1, 'fused', max_walkers=2 * n_w)
        np.random.seed(1)
        p0 = m.init_params_from_priors(n_w)
        s = DeviceEnsembleSampler(n_w, m, block=32)
        s.random_state = np.random.RandomState(5).get_state()
        lnp = m.log_posterior_batch(p0)
        nacc = np.zeros(n_w, dtype=np.int64)
        draws, states = s._draw(32)
        eng = m.engine
        res, chains, used = {}, {}, {}
        for mode in (100000, 0, -1):                      # forced on, off, the library's own rule
            eng.set_option('speculate', mode)
            before = eng.get_option('speculated_runs')
            out = eng.stretch_run(p0.copy(), lnp.copy(), *draws, nacc.copy(), store=True)
            ts = []
            for _ in range(4):
                t = time.perf_counter(); out = eng.stretch_run(p0.copy(), lnp.copy(), *draws, nacc.copy(), store=True); ts.append(time.perf_counter() - t)
            res[mode] = min(ts) * 1e3 / 32
            chains[mode] = out[2]
            used[mode] = eng.get_option('speculated_runs') > before
        same = np.array_equal(chains[100000], chains[0])
        print('%4d^2 P=%2d %4d walkers: %.4f ms/iteration with whole-iteration launches, %.4f without (x%.2f); auto -> %s %.4f; chains identical: %s'
              % (side, m.num_params, n_w, res[100000], res[0], res[0] / res[100000], 'on' if used[-1] else 'off', res[-1], same), flush=True)
        m.close()
