import sys, numpy as np
sys.path[:0]=['.','oracle','tools','tests']
import synth_field, helpers
from test_gpu_fullsize import make_model
model, fld = make_model(512, 2, 'fused', max_walkers=96)
theta = np.vstack([synth_field.draw_walkers(512, 2, 48, seed=5), synth_field.draw_walkers(512, 2, 48, seed=6, near_truth=fld['truth'])])
a = model.log_posterior_batch(theta)
for mode in (1, 2, 3):
    try:
        model.engine.set_option('rows3', mode)
        b = model.log_posterior_batch(theta)
        fin = np.isfinite(a)
        print('rows3=%d' % mode, 'finite equal', np.array_equal(np.isfinite(b), fin), 'max rel diff', np.max(np.abs(a[fin]-b[fin])/np.abs(a[fin])))
    except Exception as e:
        print('rows3=%d' % mode, 'ERR', e)
model.engine.set_option('rows3', 0)
