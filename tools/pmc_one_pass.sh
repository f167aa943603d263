#!/bin/bash
# one PMC pass over bench.py; prints per-kernel averages.  usage: _pmc1.sh "<counters>" <bench args>
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
CNT="$1"; shift
rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py "$@" --no-cpu --no-example --no-extras --steps 2 --warmup 1 --batches 1 --opt streams=1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
cc=glob.glob("$R/gpurun_out/pmc1/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(cc)):
    k=r["Kernel_Name"].split("(")[0].replace("void psfmc::","")[:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in agg:
    if "rows" in k or "cols" in k:
        print("%-42s" % k, {c: "%.3e" % (v/cnt[(k,c)]) for c,v in agg[k].items()})
PY
rm -rf $R/gpurun_out/pmc1
