#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then HBM PMC passes (separate runs).
# usage: tools/prof.sh <tag> [bench args...]      (default workload: bench.py's default, C4)
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export AL_REF_CACHE=/tmp/alcache
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}   # (set before the profiler's preloaded library initialises HIP: bench.py's own setdefault comes too late under rocprofv3)
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --f2f-pairs 0 --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $OUT/pmc_valu --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/bench_valu.json 2> $OUT/valu.err
python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
# keep only small files for the merge back (<= 64 MiB)
find $OUT -name "*.csv" -size +6M -delete
ls -la $OUT | head -20; tail -n 2 $OUT/*.err | cut -c1-200
