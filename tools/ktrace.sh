#!/bin/bash
# quick per-kernel timing: tools/ktrace.sh <tag> [bench args]
TAG=${1:-kt}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/kt_$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}   # (set before the profiler's preloaded library initialises HIP: bench.py's own setdefault comes too late under rocprofv3)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline $* > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)
for row in csv.DictReader(open(f[0])):
    print("%-60s calls %4s total %9.3f ms avg %9.1f us %5s%%" % (row["Name"][:60], row["Calls"], float(row["TotalDurationNs"])/1e6, float(row["AverageNs"])/1e3, row["Percentage"]))
PY
