#!/bin/bash
# segment chaining + big sorts: golden parity, then c4s / c4 timing
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/r2b; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_sam.py tests/test_gpu_stages.py -m gpu -x -q > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
( time python3 bench.py --config c4s --pairs 500000 --steps 3 --warmup 1 --cpu-sample-pairs 100000 ) > $O/c4s.json 2> $O/c4s.err
( time python3 bench.py --steps 3 --warmup 1 --pairs 500000 --cpu-sample-pairs 100000 ) > $O/c4.json 2> $O/c4.err
timeout 1200 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
tail -n 3 $O/*.err
