#!/bin/bash
# tools/pmc.sh <tag> "<counters>" [bench args]  -- one PMC pass (counters in their own run, kernel-trace only)
TAG=$1; CTR=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}   # (set before the profiler's preloaded library initialises HIP: bench.py's own setdefault comes too late under rocprofv3)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace -d $OUT/run --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline $* > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$OUT/run/**/*counter_collection.csv",recursive=True)
agg=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(int)
for row in csv.DictReader(open(f[0])):
    k=row["Kernel_Name"][:44]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
names=sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(46)+" ".join(n[-18:].rjust(18) for n in names))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1].values()))[:16]:
    print(k.ljust(46)+" ".join(("%.3g"%v.get(n,0)).rjust(18) for n in names))
PY
find $OUT -name "*.csv" -size +4M -delete
