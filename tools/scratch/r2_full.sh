#!/bin/bash
# full -m gpu suite + the driver's bench line (run through gpurun)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2full}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
( time timeout 1700 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 8 $O/pytest.log
( time python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} ) > $O/bench.json 2> $O/bench.err; tail -n 4 $O/bench.err; cat $O/bench.json
