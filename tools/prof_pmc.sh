#!/bin/bash
# PMC passes for the fused kernels (run on the GPU box via gpurun).
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
BENCH_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for name in ['sq1','sq2','fetch','write','tcc']:
    files = glob.glob('$OUT/%s/*/*counter_collection.csv' % name)
    if not files: print(name, 'no output'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r['Kernel_Name'].split('(')[0][-30:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k in agg:
        if 'k_rows' in k or 'k_cols' in k:
            print(name, k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
