#!/usr/bin/env python3
"""Device time per call of the raw-vector log-posterior for small batches (256^2 field),
back to back on one stream.  PSFMC_LIB selects the library build (A/B runs)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tools')]
import numpy as np
import torch
import bench
args = argparse.Namespace(size=256, sersic=1, walkers=256, backend='fused')
model, theta, fld = bench.build_problem(args, 0)
eng = model.engine
dev = torch.device('cuda', 0)
th = torch.from_numpy(theta[128:]).to(dev)          # near-truth walkers: all inside the priors
out = torch.empty(256, dtype=torch.float64, device=dev)
st = torch.cuda.Stream(dev)
res = []
for w in (11, 32, 128):
    for _ in range(20):
        eng.logpost_theta_device(w, th.data_ptr(), 0, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(300):
        eng.logpost_theta_device(w, th.data_ptr(), 0, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    res.append('W=%d %.1f us' % (w, (time.perf_counter() - t) / 300 * 1e6))
print(os.environ.get('PSFMC_LIB', 'default'), ' | '.join(res))
model.close()
