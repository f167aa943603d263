"""Shape fuzz (run by hand on a GPU box: python tests/fuzz_shapes.py <seed> <cases>): random (ny, nx) pairs of built sides,
any even sides (embedded where not built) and sides above 1024 against short partners, random fields and component sets
(test_gpu_random.random_case), the fused back end against the fp64 oracle.  Round 4: 1420 cases, no mismatch (one ill-conditioned sum, within 2e-12 of the terms' magnitudes on both back ends)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (this file lives in tests/: test infrastructure, the only place beside smoke() and the bench baseline that may use the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import psfmc_oracle as orc
import test_gpu_random as tgr
from psfmc_amd import engine
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sides = [s for s in engine.FUSED_SIDES if s <= 900]
big = [1152, 1280, 1536, 2048]
bad = 0
for i in range(n_cases):
    if i % 10 == 9:
        shape = (int(rng.choice(big)), int(rng.choice([64, 96, 128, 200, 300, 336, 512])))
        if rng.rand() < 0.5: shape = shape[::-1]
    elif i % 10 == 8:
        shape = (int(rng.randint(40, 400)) * 2, int(rng.randint(40, 400)) * 2)       # any even side: embedded if not built
    else:
        shape = (int(rng.choice(sides)), int(rng.choice(sides)))
    case = tgr.random_case(5000 + i, shape)
    if not engine.fused_supports(shape[0], shape[1], case['psfs'][0].shape):
        continue
    field = orc.make_field(case['sci'], case['ivm'], case['psfs'], case['pivms'], mask=case['mask'], mag_zp=case['zp'])
    want, imgs = orc.evaluate(field, case['comps'], case['psf_index'], raw_dtype=np.float64)
    want = want if np.isfinite(want) else -np.inf
    # a log-likelihood near zero is a cancellation of terms thousands of times larger (case 137 of seed 21: 548 out of
    # 232 434, both back ends 3e-7 from the oracle): the bound is relative to the sum of the terms' magnitudes as well
    cond = 0.0
    if np.isfinite(want):
        g = ~field.bad_px
        ivm = imgs['composite_ivm'][g]
        cond = 0.5 * float(np.abs((field.sci - imgs['convolved_model'])[g] ** 2 * ivm).sum() + np.abs(np.log(0.5 / np.pi * ivm)).sum())
    n_free = 1 if len(case['psfs']) > 1 else 0
    theta = np.full((2, n_free), float(case['psf_index']))
    model = tgr.build(case, 'fused')
    got = model.log_likelihood_batch(theta)
    ok = (got[0] == got[1]) and ((not np.isfinite(want) and got[0] == -np.inf) or (np.isfinite(want) and abs(got[0] - want) <= max(2e-10 * abs(want), 5e-12 * cond)))
    print(shape, 'transform', (int(model.engine.get_option('transform_ny')), int(model.engine.get_option('transform_nx'))), 'ok' if ok else 'MISMATCH %r %r' % (got[0], want), flush=True)
    bad += 0 if ok else 1
    model.close()
print('mismatches:', bad)
