#!/bin/bash
# file-to-file leg alone: C4 reference + PAIRS pairs on /dev/shm -> airlift-align -> SAM in /tmp; runs: default, AL_NO_RESERVE=1, default again
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-f2f}; mkdir -p $O
CFG=${CFG:-c4}
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("$CFG"); g.write_fasta("/dev/shm/sref.fa", ref)
r1, r2 = g.simulate("$CFG", ref, ${PAIRS:-4000000}, 4242)
g.write_fastq("/dev/shm/s_1.fq", r1); g.write_fastq("/dev/shm/s_2.fq", r2)
PY
i=0
for envx in ${RUNS:-"X=1" "AL_NO_RESERVE=1" "X=2"}; do
  i=$((i+1)); sleep ${SLEEP:-5}
  /usr/bin/env bash -c "time env AL_TIMING=1 $envx ${ENVX:-} $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_$i.sam /dev/shm/sref.fa /dev/shm/s_1.fq /dev/shm/s_2.fq" 2> $O/run_$i.err
  echo "== run $i ($envx)"; grep -E "^real|index build|index:|stream pipeline:|context|allocation calls|reserve|-> batches" $O/run_$i.err | cut -c1-700
done
cmp /tmp/so_1.sam /tmp/so_2.sam && echo "runs 1 and 2: identical SAM"
md5sum /tmp/so_1.sam
