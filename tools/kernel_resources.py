#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / spill / scratch / LDS / occupancy of the kernels of a HIP source
(hipcc -Rpass-analysis=kernel-resource-usage; device code only, nothing is linked).

  tools/kernel_resources.py [--filter TEXT] [--out FILE]              the library: the four parts of
                                                                      psfmc_hip.hip, compiled in parallel
  tools/kernel_resources.py --src tools/rows3_probe.hip -DSIDES=...   any other source (extra flags pass through)
"""
import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = [('VGPR', r'VGPRs'), ('AGPR', r'AGPRs'), ('SGPR', r'SGPRs'), ('spill', r'VGPR Spill'),
        ('scratch', r'ScratchSize \[bytes/lane\]'), ('occ', r'Occupancy \[waves/SIMD\]'),
        ('LDS', r'LDS Size \[bytes/block\]')]


def remarks(src, flags):
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only',
           '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + flags
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def rows(txt, flt):
    out = []
    for blk in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
        name = blk.split('\n')[0].strip().split(' ')[0]
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r'^void ', '', re.sub(r'\(.*', '', dn))
        if flt not in dn:
            continue
        vals = []
        for label, key in KEYS:
            m = re.search(key + r': (\d+)', blk)
            vals.append('%s %s' % (label, m.group(1) if m else '?'))
        out.append('%-78s %s' % (dn[:78], '  '.join(vals)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--src')
    ap.add_argument('--filter', default='')
    ap.add_argument('--out')
    args, flags = ap.parse_known_args()
    if args.src:
        lines = rows(remarks(args.src, flags), args.filter)
    else:
        src = os.path.join(ROOT, 'psfmc_amd', 'csrc', 'psfmc_hip.hip')
        with ThreadPoolExecutor(4) as ex:
            txts = list(ex.map(lambda k: remarks(src, ['-DPSFMC_NPARTS=4', '-DPSFMC_PART=%d' % k] + flags), range(4)))
        lines = sorted(set(l for t in txts for l in rows(t, args.filter)))
    text = '\n'.join(lines)
    print(text)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
