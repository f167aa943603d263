#!/usr/bin/env python3
"""How the two lanes of a batch overlap, from a rocprofv3 --kernel-trace CSV: per queue the kernels'
durations and the gaps between consecutive kernels, and the share of the batch's wall time with
0 / 1 / 2+ pipeline kernels running.
usage: lane_timeline.py <kernel_trace.csv> [max batches]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'psfmc::k_' in r['Kernel_Name']]
ev = []
for r in rows:
    name = r['Kernel_Name'].split('psfmc::')[1].split('<')[0].split('(')[0]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], name))
ev.sort()
pipe = [e for e in ev if e[3] in ('k_rows_fwd', 'k_cols', 'k_cols3', 'k_cols3g', 'k_cols3f', 'k_rows_inv', 'k_rows3_fwd', 'k_rows3_inv')]
# batches are separated by k_theta_prep
starts = [e[0] for e in ev if e[3] == 'k_theta_prep']
if len(starts) < 4:
    sys.exit('not enough batches')
lo, hi = starts[len(starts) // 2], starts[len(starts) // 2 + 1]          # one batch in the middle
batch = [e for e in pipe if lo <= e[0] < hi]
t0, t1 = min(e[0] for e in batch), max(e[1] for e in batch)
print('batch: %d pipeline kernels, %.1f us wall' % (len(batch), (t1 - t0) / 1e3))
byq = collections.defaultdict(list)
for e in batch:
    byq[e[2]].append(e)
for q, es in sorted(byq.items()):
    dur = collections.defaultdict(list)
    gaps = []
    for i, e in enumerate(es):
        dur[e[3]].append((e[1] - e[0]) / 1e3)
        if i:
            gaps.append((e[0] - es[i - 1][1]) / 1e3)
    print('queue %s: %d kernels; ' % (q, len(es)) + '  '.join('%s %.1f us avg' % (k, sum(v) / len(v)) for k, v in dur.items()) +
          '; gap between consecutive kernels %.1f us avg (max %.1f)' % (sum(gaps) / max(len(gaps), 1), max(gaps or [0])))
# concurrency profile
pts = sorted([(e[0], 1) for e in batch] + [(e[1], -1) for e in batch])
level, last, share = 0, t0, collections.Counter()
for t, d in pts:
    share[min(level, 2)] += t - last
    level += d
    last = t
tot = float(t1 - t0)
print('share of wall time with 0 / 1 / 2+ kernels running: %.1f %% / %.1f %% / %.1f %%' %
      (100 * share[0] / tot, 100 * share[1] / tot, 100 * share[2] / tot))
# which pairs overlap
pair = collections.Counter()
for i, a in enumerate(batch):
    for b in batch[i + 1:]:
        if b[0] >= a[1]:
            break
        if a[2] != b[2]:
            ov = min(a[1], b[1]) - b[0]
            if ov > 0:
                pair[tuple(sorted((a[3], b[3])))] += ov
for k, v in pair.most_common():
    print('  overlap %-24s %.1f %% of wall' % ('%s + %s' % k, 100 * v / tot))
