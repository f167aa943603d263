#!/usr/bin/env python3
"""Print VGPR / spill / LDS / occupancy of every kernel in psfmc_hip.hip
(hipcc -Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'psfmc_amd', 'csrc', 'psfmc_hip.hip')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-c', src,
       '-o', '/tmp/psfmc_res.o', '-Rpass-analysis=kernel-resource-usage']
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ''
KEYS = [('VGPR', r'VGPRs'), ('AGPR', r'AGPRs'), ('SGPR', r'SGPRs'), ('spill', r'VGPR Spill'),
        ('scratch', r'ScratchSize \[bytes/lane\]'), ('occ', r'Occupancy \[waves/SIMD\]'),
        ('LDS', r'LDS Size \[bytes/block\]')]
for blk in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = blk.split('\n')[0].strip()
    dn = subprocess.run(['c++filt', name], capture_output=True,
                        text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn)
    if flt not in dn:
        continue
    vals = []
    for label, key in KEYS:
        m = re.search(key + r': (\d+)', blk)
        vals.append('%s %s' % (label, m.group(1) if m else '?'))
    print('%-60s %s' % (dn[:60], '  '.join(vals)))
