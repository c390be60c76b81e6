#!/bin/bash
# kernel timeline of one resident step: tools/tl.sh <tag> [min_us]
TAG=${1:-tl}; MINUS=${2:-300}
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/tl_$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}   # (set before the profiler's preloaded library initialises HIP: bench.py's own setdefault comes too late under rocprofv3)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline --f2f-pairs 0 --steps 1 --warmup 1 $TLARGS > $OUT/bench.json 2> $OUT/err.txt
python3 $REPO/tools/timeline.py $OUT/trace $MINUS > $OUT/timeline.txt
find $OUT -name "*.csv" -size +6M -delete
