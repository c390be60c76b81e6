#!/bin/bash
# the file-to-file leg ALONE on the bench's own data (bench.make_workload, seed 20261002): is the bench's slower leg the data or the environment?
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<PY
import sys, os; sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
import gen_synth as g, bench as B, numpy as np
ref = g.build_reference("c4"); g.write_fasta("/dev/shm/sref.fa", ref)
arr = B.make_workload("c4", 0, 6250000, 150, 20261002, ref, None)
for m, fn in ((0, "/dev/shm/b_1.fq"), (1, "/dev/shm/b_2.fq")): B.write_fastq_fast(fn, arr, m, 0)
PY
IFS=";" read -ra SETS <<< "${1:-X=1;X=2}"; i=0; for e in "${SETS[@]}"; do i=$((i+1)); sleep 4; echo "== $e"; ( time timeout 120 env AL_TIMING=1 $e $REPO/airlift_amd/bin/airlift-align -ax sr -t 32 -o /tmp/sb_$i.sam /dev/shm/sref.fa /dev/shm/b_1.fq /dev/shm/b_2.fq ) 2>&1 | grep -E "stream pipeline: 1 lane|pipeline lane 0|^real" | cut -c1-330; done
