#!/bin/bash
# Measurement aid for the paired pipeline (GPU box): time a batch with only some of the roles running
# (results are wrong unless all run).  usage: tools/pair_roles.sh <tag> <size> <sersic> <walkers>
TAG=$1; N=$2; S=$3; W=$4
R=${GRAFT_REPO_ROOT:-.}
for roles in 3 1 2 5 9 0; do
  python3 $R/bench.py --size $N --sersic $S --walkers $W --no-cpu --no-example --no-extras --steps 5 --warmup 2 --step-ms 60 --opt pair_roles=$roles 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d.get('kernels',[]) if 'pair' in x['kernel']]
print('roles=$roles  %9.0f evals/s   per-batch %.3f ms' % (d['value'], d['ms_per_step']/d['config']['batches_per_step']), [(x['kernel'], round(x['avg_ms']*1e3,1)) for x in k])
" | tee -a $R/gpurun_out/${TAG}_roles.txt
done
