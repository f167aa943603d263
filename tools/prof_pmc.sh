#!/bin/bash
# rocprofv3 PMC passes over bench.py (run on the GPU box via gpurun).  Counters are
# collected in their own runs (no trace domains), FETCH_SIZE and WRITE_SIZE apart
# (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
BENCH_ARGS="$*"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
# request-size breakdown at the L2's memory side: an exact byte count to set beside the
# guide's FETCH_SIZE x 2 rule (which assumes every read request is a 128-B one)
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
args=""
for name in sq1 sq2 fetch write rdreq wrreq; do
  f=$(ls $OUT/$name/*/*counter_collection.csv 2>/dev/null | head -1)
  args="$args $name=${f:-missing}"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $args
