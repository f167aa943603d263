#!/usr/bin/env python3
"""Instruction mix of selected kernels from the gfx950 assembly.
Usage: tools/asm_stats.py <pattern> [...]   (patterns match mangled names)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'psfmc_amd', 'csrc', 'psfmc_hip.hip')
out = '/tmp/psfmc_asm.s'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S',
                '--cuda-device-only', '-o', out, src], capture_output=True)
lines = open(out).read().split('\n')
pats = sys.argv[1:] or ['k_rows_invILi256', 'k_rows_fwdILi256ELb0', 'k_colsILi256ELb1']
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_ZN5psfmc[^ ]*:', l)]
for idx, (i, name) in enumerate(starts):
    if not any(p in name for p in pats):
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
    ops = collections.Counter()
    for l in lines[i + 1:end]:
        m = re.match(r'\s+([a-z][a-z_0-9]+)\s', l)
        if m:
            ops[m.group(1)] += 1
    tot = sum(ops.values())
    f64 = sum(v for k, v in ops.items() if 'f64' in k)
    trans = sum(v for k, v in ops.items() if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)', k))
    print('%s\n   total %d  f64 %d  transcendental %d  ds %d  global/buffer %d  s_waitcnt %d  s_barrier %d'
          % (name[:70], tot, f64, trans, sum(v for k, v in ops.items() if k.startswith('ds_')),
             sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))),
             ops['s_waitcnt'], ops['s_barrier']))
    print('   ' + ', '.join('%s:%d' % kv for kv in ops.most_common(18)))
