#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box and stage the summaries for
# profiles/ (run through gpurun; results come back under gpurun_out/profiles_new and are
# copied into profiles/ by hand).
# usage: tools/collect_profiles.sh <round tag, e.g. r2> [configs: any of 256 512 1024 200 300, default the first three]
# Configs (BASELINE.json / SURVEY.md section 8(d)):
#   256   headline: 256^2, 1 PS + 1 Sersic, 4096 walkers per batch
#   512   config 3: 512^2, 1 PS + 2 Sersic, 1024 walkers
#   1024  config 4's per-GPU share: 1024^2, 1 PS + 4 Sersic, 256 walkers
set -e -o pipefail
TAG=${1:-r2}; shift || true
CONFIGS=${*:-256 512 1024}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
export TMPDIR=/tmp
cp $R/profiles/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null || true
for N in $CONFIGS; do
  case $N in
    # CH = walkers per launch of the bench's own kernel-profile pass (one pass in flight, the library's
    # own pass size): batch / passes
    256)  ARGS="--size 256 --sersic 1 --walkers 4096"; CH=110.7027 ;;     # 4096 / 37
    512)  ARGS="--size 512 --sersic 2 --walkers 1024"; CH=23.8140 ;;      # 1024 / 43
    1024) ARGS="--size 1024 --sersic 4 --walkers 256"; CH=5.9535 ;;       # 256 / 43
    200)  ARGS="--size 200 --sersic 1 --walkers 4096"; CH=178.0870 ;;     # 4096 / 23
    300)  ARGS="--size 300 --sersic 1 --walkers 2048"; CH=78.7692 ;;      # 2048 / 26
    2048) ARGS="--size 2048 --sersic 1 --walkers 64"; CH=1 ;;              # one walker per pass
    *)    ARGS="--size $N --sersic 1 --walkers 1024"; CH=0 ;;              # any other side: CH from the bench's own kernel pass (below)
  esac
  COMMON="$ARGS --no-cpu --no-example --no-extras"
  cd /tmp
  # (1) kernel trace of the bench command as it runs by default (two passes in flight)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_$N -- python3 $R/bench.py $COMMON --steps 5 --warmup 2 > $OUT/${TAG}_bench_${N}_under_rocprof.json 2>/dev/null
  cp $R/gpurun_out/${TAG}_stats_$N/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_$N.csv
  if [ "$CH" = "0" ]; then
    CH=$(python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['kernels'][0]['walkers_per_launch'])" $OUT/${TAG}_bench_${N}_under_rocprof.json)
  fi
  # (2) the same with one pass in flight: every kernel runs alone, which is what the
  #     bench's in-library HIP-event pass (roofline object) measures
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_${N}_s1 -- python3 $R/bench.py $COMMON --steps 5 --warmup 2 --opt streams=1 > /dev/null 2>&1
  cp $R/gpurun_out/${TAG}_stats_${N}_s1/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_${N}_streams1.csv
  # (3) PMC passes (own runs, no trace domains), one pass in flight at the library's own pass size: the
  #     launches the bench's roofline object times (CH walkers per launch on average)
  cd $R
  tools/prof_pmc.sh ${TAG}_pmc_$N $COMMON --batches 1 --opt streams=1 > $OUT/${TAG}_pmc_${N}_summary.txt 2>&1
  python3 tools/pmc_to_json.py gpurun_out/${TAG}_pmc_$N $N $CH $OUT/pmc_traffic.json > /dev/null
  echo "config $N done"
done
# the plain bench lines (traffic from the PMC table just written)
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
cd $R
for N in $CONFIGS; do
  case $N in
    256)  python3 bench.py > $OUT/${TAG}_bench.json 2>$OUT/${TAG}_bench.err ;;   # (the default line, CPU legs included)
    512)  python3 bench.py --size 512 --sersic 2 --walkers 1024 --no-example --cpu-seconds 6 > $OUT/${TAG}_bench_512.json 2>$OUT/${TAG}_bench_512.err ;;
    1024) python3 bench.py --size 1024 --sersic 4 --walkers 256 --no-example --cpu-seconds 6 --cpu-procs 0 > $OUT/${TAG}_bench_1024.json 2>$OUT/${TAG}_bench_1024.err ;;
    200)  python3 bench.py --size 200 --sersic 1 --walkers 4096 --no-example --cpu-seconds 6 --cpu-procs 0 > $OUT/${TAG}_bench_200.json 2>$OUT/${TAG}_bench_200.err ;;
    300)  python3 bench.py --size 300 --sersic 1 --walkers 2048 --no-example --cpu-seconds 6 --cpu-procs 0 > $OUT/${TAG}_bench_300.json 2>$OUT/${TAG}_bench_300.err ;;
    2048) python3 bench.py --size 2048 --sersic 1 --walkers 64 --no-example --cpu-seconds 6 --cpu-procs 0 > $OUT/${TAG}_bench_2048.json 2>$OUT/${TAG}_bench_2048.err ;;
    *)    python3 bench.py --size $N --sersic 1 --walkers 1024 --no-example --cpu-seconds 6 --cpu-procs 0 > $OUT/${TAG}_bench_$N.json 2>$OUT/${TAG}_bench_$N.err ;;
  esac
done
tail -c 600 $OUT/${TAG}_bench*.json
if [ -n "$PSFMC_PROFILES_SIZES_ONLY" ]; then
  rm -rf $R/gpurun_out/${TAG}_stats_* $R/gpurun_out/${TAG}_pmc_*
  exit 0
fi
# BASELINE config 5's per-GPU share (8 independent 256^2 fields x 256 walkers each: in one shared
# context, and with a context per field), the
# small-ensemble timeline and the device-resident sampler's trace
python3 bench.py --fields 8 --walkers 256 --no-cpu --no-example > $OUT/${TAG}_bench_fields8.json 2>/dev/null || true
python3 bench.py --fields 8 --walkers 256 --no-cpu --no-example --fields-merge 0 > $OUT/${TAG}_bench_fields8_own_contexts.json 2>/dev/null || true
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_small -o small -- python3 $R/tools/trace_small.py > /dev/null 2>&1 || true
python3 $R/tools/trace_small.py --analyse $R/gpurun_out/${TAG}_small/small_kernel_trace.csv > $OUT/${TAG}_small_ensemble_timeline.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_sampler -o s256 -- python3 $R/tests/profile_sampler.py 256 300 > $OUT/${TAG}_sampler_256.txt 2>/dev/null || true
cp $R/gpurun_out/${TAG}_sampler/s256_kernel_stats.csv $OUT/${TAG}_sampler_256_kernel_stats.csv 2>/dev/null || true
cd $R
cat $OUT/${TAG}_small_ensemble_timeline.txt
# only the summaries travel back (gpurun merges at most 64 MiB; the raw traces are hundreds)
rm -rf $R/gpurun_out/${TAG}_stats_* $R/gpurun_out/${TAG}_pmc_* $R/gpurun_out/${TAG}_small $R/gpurun_out/${TAG}_sampler
