#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 PMC passes of tools/prof_pmc.sh.
usage: pmc_summary.py <pass name>=<counter_collection.csv> ..."""
import collections
import csv
import sys


def hot(k):
    # the hot-path instantiations: rasterising forward rows, convolving columns, inverse rows
    return (k.startswith('k_rows_fwd<') and ', false' in k.split('<')[1][:12]) or k.startswith('k_rows_inv<') or \
        (k.startswith('k_cols') and ', true' in k) or k.startswith(('k_theta_prep', 'k_finish'))


for arg in sys.argv[1:]:
    name, path = arg.split('=', 1)
    try:
        rows = list(csv.DictReader(open(path)))
    except (IOError, OSError):
        print(name, 'no output')
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void psfmc::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k in sorted(agg):
        if hot(k):
            print(name, k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in agg[k].items()})
