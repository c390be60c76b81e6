#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2f}; mkdir -p $O
cd $REPO
( time timeout 900 python3 -m pytest tests/test_gpu_sam.py -m gpu -x -q -k "cli_sam or above_the_limit or stage3" ) > $O/pytest1.log 2>&1; tail -15 $O/pytest1.log
ls gpurun_out/samdiff* 2>/dev/null | head
