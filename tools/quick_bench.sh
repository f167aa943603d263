#!/bin/bash
# Quick A/B figures on the GPU box: the three single-GPU configs without the CPU legs.
# usage: tools/quick_bench.sh <tag> [extra bench args]
TAG=${1:-q}; shift || true
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
for cfg in "--size 256 --sersic 1 --walkers 4096" "--size 512 --sersic 2 --walkers 1024" "--size 1024 --sersic 4 --walkers 256"; do
  python3 $R/bench.py $cfg --no-cpu --no-example --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s %10.0f evals/s  step-frac %s' % (d['config']['workload'][:60], d['value'], d.get('roofline_step',{}).get('frac')))
for k in d.get('kernels',[]): print('    %-28s %7.1f us  %6.0f GB/s' % (k['kernel'], k['avg_ms']*1e3, k['GBps']))
print('    small', {k: round(v['us_per_call'],1) for k,v in d.get('small_ensembles',{}).items()})
" | tee -a $R/gpurun_out/${TAG}_quick.txt
done
