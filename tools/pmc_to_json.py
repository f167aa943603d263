#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) into
profiles/pmc_traffic.json: measured HBM bytes per walker for each fused kernel.
Correction per /opt/skills/guides/MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE
is exact for 16 B/lane streaming stores.  Both counters are in KiB.
usage: pmc_to_json.py <pmc dir> <image side> <walkers per launch> <out json>"""
import collections
import csv
import glob
import json
import os
import sys

pmc_dir, side, walkers, out = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]


def per_kernel(name, counter):
    files = glob.glob(os.path.join(pmc_dir, name, '*', '*counter_collection.csv'))
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void psfmc::', '').strip()
        agg[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return {k: agg[k] / cnt[k] for k in agg}


def optional(name, counter):
    try:
        return per_kernel(name, counter)
    except (IndexError, IOError, OSError):
        return {}


fetch = per_kernel('fetch', 'FETCH_SIZE')
write = per_kernel('write', 'WRITE_SIZE')
rd = {c: optional('rdreq', 'TCC_EA0_RDREQ' + c + '_sum') for c in ('', '_32B', '_64B', '_128B')}
wr = {c: optional('wrreq', 'TCC_EA0_WRREQ' + c + '_sum') for c in ('', '_64B')}
valu = optional('sq2', 'SQ_INSTS_VALU')          # vector wave-instructions per launch
table = {}
for k in sorted(set(fetch) | set(write)):
    if not any(s in k for s in ('k_rows_fwd<%d, false' % side, 'k_cols<%d, true' % side, 'k_cols3<%d, true' % side,
                                'k_cols3g<%d, true' % side, 'k_cols3f<%d, true' % side, 'k_rows_inv<%d' % side,
                                'k_rows3_fwd<%d, false' % side, 'k_rows3_inv<%d' % side)):
        continue
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    table[k] = {'FETCH_SIZE_KiB_per_launch': f_kib, 'WRITE_SIZE_KiB_per_launch': w_kib,
                'walkers_per_launch': walkers,
                'hbm_bytes_per_walker': (2.0 * f_kib + w_kib) * 1024.0 / walkers}
    if k in valu:
        table[k]['valu_wave_instructions_per_walker'] = valu[k] / walkers
    if k in rd['']:
        # exact request sizes (cross-check of the x2 rule): requests not tallied as 32- / 64- /
        # 128-byte ones are taken as 64-byte
        n, n32, n64, n128 = (rd[c].get(k, 0.0) for c in ('', '_32B', '_64B', '_128B'))
        other = max(n - n32 - n64 - n128, 0.0) if (n64 or n128) else 0.0
        rbytes = 32 * n32 + 64 * (n64 + other) + 128 * n128 if (n64 or n128) else None
        table[k]['read_requests_per_launch'] = {'all': n, '32B': n32, '64B': n64, '128B': n128}
        if rbytes is not None:
            table[k]['exact_read_bytes_per_walker'] = rbytes / walkers
    if k in wr['']:
        nw, nw64 = wr[''].get(k, 0.0), wr['_64B'].get(k, 0.0)
        table[k]['write_requests_per_launch'] = {'all': nw, '64B': nw64}
        table[k]['exact_write_bytes_per_walker'] = (64 * nw64 + 32 * max(nw - nw64, 0.0)) / walkers
try:
    full = json.load(open(out))
except (IOError, ValueError):
    full = {}
full['%dx%d' % (side, side)] = table
full['_note'] = ('hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled per '
                 'MI355X_MICROARCH.md (gfx950 counts 128-B read requests as 64 B)')
json.dump(full, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(table, indent=1))
