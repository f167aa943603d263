#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box and copy the summaries into
# profiles/ (run through gpurun; profiles/ is merged back via gpurun_out/profiles_new).
# usage: tools/collect_profiles.sh <round tag, e.g. r1>
set -e -o pipefail
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace of the default bench command (two passes in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_default -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/${TAG}_bench_default_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/${TAG}_stats_default/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_default.csv
# (2) the same with one pass in flight: every kernel runs alone, which is what the
#     bench's in-library HIP-event pass (roofline object) measures
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_streams1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --opt streams=1 > $OUT/${TAG}_bench_streams1_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/${TAG}_stats_streams1/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_streams1.csv
# (3) hipFFT cross-check back end
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_hipfft -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --backend hipfft > /dev/null 2>&1
cp $R/gpurun_out/${TAG}_stats_hipfft/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_hipfft_backend.csv
# (4) PMC passes (own runs, no trace domains), 128 walkers per launch
cd $R
tools/prof_pmc.sh ${TAG}_pmc --opt streams=1 --chunk 128 > $OUT/${TAG}_pmc_summary.txt 2>&1
for n in sq1 sq2 fetch write; do cp gpurun_out/${TAG}_pmc/$n/*/*counter_collection.csv $OUT/${TAG}_pmc_${n}_counter_collection.csv; done
python3 tools/pmc_to_json.py gpurun_out/${TAG}_pmc 256 128 $OUT/pmc_traffic.json > /dev/null
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
# (5) the plain bench line (traffic now comes from the PMC table just written)
python3 bench.py > $OUT/${TAG}_bench.json 2>/dev/null
tail -c 400 $OUT/${TAG}_bench.json
# (6) device-resident sampler, 256 walkers: kernel stats and the timeline of one iteration
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_sampler -o s256 -- python3 $R/tests/profile_sampler.py 256 300 > $OUT/${TAG}_sampler_256.txt 2>/dev/null
cp $R/gpurun_out/${TAG}_sampler/s256_kernel_stats.csv $OUT/${TAG}_sampler_256_kernel_stats.csv
python3 - $R/gpurun_out/${TAG}_sampler/s256_kernel_trace.csv > $OUT/${TAG}_sampler_256_trace.txt <<'PY'
import csv, sys
tr = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
mid = tr[len(tr) // 2:len(tr) // 2 + 16]
t0 = int(mid[0]['Start_Timestamp'])
print('start_us  duration_us  kernel   (16 consecutive launches from the middle of the run)')
for r in mid:
    print('%8.1f %8.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3,
                               (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:60]))
PY
cd $R
