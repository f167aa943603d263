#!/usr/bin/env python3
"""GPU diagnostic: error of each backend vs the reference / fp64 oracle for every
golden case (max and which vector).  Run on the GPU box."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/oracle', ROOT + '/tools', ROOT + '/tests']
import helpers  # noqa: E402

for name in ['example', 'synth256', 'synth128x2', 'edge']:
    case = helpers.load_case(name)
    for backend in ['hipfft', 'fused']:
        with tempfile.TemporaryDirectory() as tmp:
            model = helpers.build_model(name, case, tmp, backend=backend, max_walkers=256)
            fin = np.isfinite(case['lnprior'])
            ll = model.log_likelihood_batch(case['params'][fin])
            ref = case['loglike_f64'][fin]
            ok = np.isfinite(ref)
            same_inf = np.array_equal(np.isfinite(ll), ok)
            rel = np.abs(ll[ok] - ref[ok]) / np.abs(ref[ok])
            ab = np.abs(ll[ok] - ref[ok])
            j = np.argmax(rel)
            small = np.abs(ref[ok]) < 5e5
            print('%-11s %-7s inf-match %s  max rel %.2e (ll=%.6g abs %.2e)  max abs near-mode %.2e  median rel %.1e'
                  % (name, backend, same_inf, rel[j], ref[ok][j], ab[j],
                     ab[small].max() if small.any() else np.nan, np.median(rel)))
            model.close()
