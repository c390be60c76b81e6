#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2c}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_sam.py tests/test_gpu_stages.py -m gpu -x -q > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
( time python3 bench.py --steps 3 --warmup 1 --pairs ${PAIRS:-500000} --cpu-sample-pairs 100000 ) > $O/c4.json 2> $O/c4.err
timeout 1200 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 --pairs ${PAIRS:-500000} > $O/c4_trace.json 2> $O/c4_trace.err
find $O -name "*.csv" -size +4M -delete
tail -n 3 $O/c4.err
