#!/bin/bash
# Quick A/B figures on the GPU box: single-GPU configs without the CPU legs.
# usage: tools/quick_bench.sh <tag> [extra bench args]     (CONFIGS="256:1:4096 512:2:1024" to choose)
TAG=${1:-q}; shift || true
R=${GRAFT_REPO_ROOT:-.}
CONFIGS=${CONFIGS:-"256:1:4096 512:2:1024 1024:4:256"}
mkdir -p $R/gpurun_out
for cfg in $CONFIGS; do
  IFS=: read size sersic walkers <<< "$cfg"
  python3 $R/bench.py --size $size --sersic $sersic --walkers $walkers --no-cpu --no-example --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=d['config']['image']; t4=4*2*(n//2+1)*n*16
print('%-60s %10.0f evals/s  designed-bytes rate %.0f GB/s  step-frac(measured) %s' % (d['config']['workload'][:60], d['value'], d['value']*t4/1e9, d.get('roofline_step',{}).get('frac')))
for k in d.get('kernels',[]): print('    %-28s %7.1f us  %6.0f GB/s' % (k['kernel'], k['avg_ms']*1e3, k['GBps']))
print('    small', {k: round(v['us_per_call'],1) for k,v in d.get('small_ensembles',{}).items()})
" | tee -a $R/gpurun_out/${TAG}_quick.txt
done
