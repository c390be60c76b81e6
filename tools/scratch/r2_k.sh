#!/bin/bash
# steady-state rate of the drop-in on C4 against the mini-batch size (-K): PAIRS pairs from files, one lane
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2k}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("c4"); g.write_fasta("/tmp/c4ref.fa", ref)
r1, r2 = g.simulate("c4", ref, ${PAIRS:-4000000}, 4242)
g.write_fastq("/tmp/c4_1.fq", r1); g.write_fastq("/tmp/c4_2.fq", r2)
PY
for k in ${KS:-50M 150M 300M}; do
  /usr/bin/env bash -c "time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-48} -K $k -o /tmp/o_$k.sam /tmp/c4ref.fa /tmp/c4_1.fq /tmp/c4_2.fq" 2> $O/cli_$k.err
  echo "-K $k"; grep -E "real|index build|lane 0" $O/cli_$k.err | cut -c1-400
done
cmp /tmp/o_50M.sam /tmp/o_300M.sam && echo "identical"
