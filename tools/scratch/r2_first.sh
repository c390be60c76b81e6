#!/bin/bash
# round-2 first look: new workloads on the round-1 kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/r2a; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
nproc > $O/host.txt; free -g >> $O/host.txt; df -h /tmp >> $O/host.txt
( time python3 bench.py --config c4s --pairs 500000 --steps 3 --warmup 1 --cpu-sample-pairs 100000 ) > $O/c4s.json 2> $O/c4s.err
( time python3 bench.py --steps 5 --warmup 2 ) > $O/c4.json 2> $O/c4.err
( time python3 bench.py --config c3 --steps 5 --warmup 2 ) > $O/c3.json 2> $O/c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/c4_trace.json 2> $O/c4_trace.err
find $O -name "*.csv" -size +8M -delete
tail -3 $O/*.err
