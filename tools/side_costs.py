#!/usr/bin/env python3
"""Per-side kernel costs of the fused back end on this GPU: for every built transform side (square
synthetic field, 1 PointSource + 1 Sersic) the device time of the three kernels per pass (the library's
own HIP events, one pass in flight) as ns per pixel per walker, and the whole-step rate.  The embedding
of unbuilt image sides (psfmc_hip.hip embed_axis) ranks candidate transform sides with this table:
a larger side with better kernels can be cheaper than the smallest one that fits.

usage (GPU box): python3 tools/side_costs.py [side ...] > gpurun_out/side_costs.jsonl"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import numpy as np
    import torch
    import bench
    from psfmc_amd import engine as engine_mod
    sides = [int(s) for s in sys.argv[1:]] or list(engine_mod.FUSED_SIDES)
    dev = torch.device('cuda:0')
    for n in sides:
        t_bytes = 2 * (n // 2 + 1) * n * 16
        walkers = int(max(8, min(4096, 6 * 110e6 // t_bytes)))
        args = argparse.Namespace(size=n, sersic=1, walkers=walkers, backend='fused')
        model, theta, fld = bench.build_problem(args, 0)
        eng = model.engine
        theta_dev = torch.as_tensor(theta, device=dev)
        out = torch.empty(walkers, dtype=torch.float64, device=dev)
        stream = torch.cuda.Stream(dev)

        def one_batch():
            eng.logpost_theta_device(walkers, theta_dev.data_ptr(), 0, out.data_ptr(), stream.cuda_stream)
        for _ in range(2):
            one_batch()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(6):
            one_batch()
        torch.cuda.synchronize(dev)
        rate = walkers * 6 / (time.perf_counter() - t0)
        prof = bench.kernel_profile(eng, args, one_batch, torch, dev, 6)
        rec = {'side': n, 'walkers': walkers, 'evals_per_s': rate, 'ns_per_pixel_step': 1e9 / rate / (n * n)}
        for k in prof:
            name = k['kernel']                     # k_rows_fwd / k_rows3_fwd, k_rows_inv / k_rows3_inv, k_cols*
            key = 'rows_fwd' if '_fwd<' in name else ('rows_inv' if '_inv<' in name else 'cols')
            rec[key + '_ns_per_pixel'] = k['avg_ms'] * 1e6 / k['walkers_per_launch'] / (n * n)
            rec[key + '_kernel'] = k['kernel']
        print(json.dumps(rec), flush=True)
        del model, eng, theta_dev, out
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
