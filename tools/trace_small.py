#!/usr/bin/env python3
"""Small-ensemble timeline helper: call the raw-vector log-posterior for a few batch sizes
back to back (run under `rocprofv3 --kernel-trace`), or, with --analyse <kernel_trace.csv>,
print per batch size the average duration of every kernel of a call and the gaps between them.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/trace_small.py
       python3 tools/trace_small.py --analyse OUT/.../*_kernel_trace.csv"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tools')]
SIZES, REPS = (11, 19, 64, 128), 40


def run():
    import argparse
    import numpy as np
    import torch
    import bench
    args = argparse.Namespace(size=256, sersic=1, walkers=256, backend='fused')
    model, theta, fld = bench.build_problem(args, 0)
    eng = model.engine
    dev = torch.device('cuda', 0)
    th = torch.from_numpy(theta).to(dev)
    out = torch.empty(256, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(dev)
    for w in SIZES:
        for _ in range(REPS):
            eng.logpost_theta_device(w, th.data_ptr(), 0, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(dev)
    model.close()


def analyse(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    def hot(name):          # the five kernels of a call (not the set-up instances of the same templates)
        return ('k_theta_prep' in name or 'k_finish_posterior' in name or 'k_rows_inv' in name or
                ('k_rows_fwd' in name and ', false' in name) or ('k_cols' in name and ', true' in name))
    rows = [r for r in rows if hot(r['Kernel_Name'])]
    calls, cur = [], []
    for r in rows:
        if 'k_theta_prep' in r['Kernel_Name'] and cur:
            calls.append(cur)
            cur = []
        cur.append(r)
    calls.append(cur)
    calls = [c for c in calls if len(c) == 5]
    per = len(calls) // len(SIZES)
    for i, w in enumerate(SIZES):
        grp = calls[i * per + per // 2:(i + 1) * per]            # second half: warmed up
        names = ['theta_prep', 'rows_fwd', 'cols', 'rows_inv', 'finish']
        dur = [sum((int(c[j]['End_Timestamp']) - int(c[j]['Start_Timestamp'])) for c in grp) / len(grp) / 1e3
               for j in range(5)]
        gap = [sum((int(c[j + 1]['Start_Timestamp']) - int(c[j]['End_Timestamp'])) for c in grp) / len(grp) / 1e3
               for j in range(4)]
        span = sum((int(c[4]['End_Timestamp']) - int(c[0]['Start_Timestamp'])) for c in grp) / len(grp) / 1e3
        period = (int(grp[-1][0]['Start_Timestamp']) - int(grp[0][0]['Start_Timestamp'])) / (len(grp) - 1) / 1e3
        print('W=%3d  call span %.1f us, period %.1f us | ' % (w, span, period) +
              '  '.join('%s %.1f' % (n, d) for n, d in zip(names, dur)) +
              ' | gaps ' + ' '.join('%.1f' % g for g in gap))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--analyse':
        analyse(sys.argv[2])
    else:
        run()
