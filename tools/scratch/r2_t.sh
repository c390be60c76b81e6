#!/bin/bash
# parity tests that exercise the seed/sort/chain paths, then the traced C4 bench (run through gpurun)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2t}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
( time timeout 1500 python3 -m pytest tests/test_gpu_sam.py tests/test_gpu_stages.py -m gpu -x -q ${PYTEST_ARGS:-} ) > $O/pytest1.log 2>&1; tail -n 6 $O/pytest1.log
TAG=${TAG:-r2t} PAIRS=${PAIRS:-1000000} bash tools/r2_quick.sh
if [ -n "${CONFIGS:-}" ]; then cd $REPO; timeout 1200 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q > $O/pytest2.log 2>&1; tail -n 3 $O/pytest2.log; fi
