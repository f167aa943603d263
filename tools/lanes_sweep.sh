#!/bin/bash
R=$GRAFT_REPO_ROOT
run() { # size sersic walkers streams chunk
  python3 $R/bench.py --size $1 --sersic $2 --walkers $3 --no-cpu --no-example --no-extras --steps 10 --warmup 2 --opt streams=$4 --chunk $5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('size $1 streams $4 chunk $5 : %.0f evals/s' % d['value'])"
}
for cfg in "512 2 1024 24 16 12" "1024 4 256 6 4 3" "256 1 4096 112 76 56"; do
  set -- $cfg
  for st in 2 3 4; do
    for ch in $4 $5 $6; do run $1 $2 $3 $st $ch; done
  done
done
