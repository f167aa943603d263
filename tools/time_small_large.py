"""Device time per call of the raw-vector log-posterior for small batches on a transform above 256 pixels per row
(the power-table rasteriser: entries formed in the row waves, or k_pow_tables).  usage: python3 tools/time_small_large.py [side] [sersic]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tools')]
import numpy as np
import torch
import bench
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
args = argparse.Namespace(size=side, sersic=ns, walkers=256, backend='fused')
model, theta, fld = bench.build_problem(args, 0)
eng = model.engine
dev = torch.device('cuda', 0)
th = torch.from_numpy(theta[128:]).to(dev)
out = torch.empty(256, dtype=torch.float64, device=dev)
st = torch.cuda.Stream(dev)
res = []
for w in (8, 16, 24, 32, 48, 64, 96, 128):
    for _ in range(10):
        eng.logpost_theta_device(w, th.data_ptr(), 0, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(100):
        eng.logpost_theta_device(w, th.data_ptr(), 0, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    res.append('W=%d %.1f' % (w, (time.perf_counter() - t) / 100 * 1e6))
print(side, ns, 'us per call:', ' | '.join(res))
model.close()
