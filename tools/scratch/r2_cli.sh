#!/bin/bash
# steady-state rate of the drop-in on C4: 2 M pairs from files, one lane vs two lanes on the GPU, vs the reference
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2cli}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("c4"); g.write_fasta("/tmp/c4ref.fa", ref)
r1, r2 = g.simulate("c4", ref, ${PAIRS:-2000000}, 4242)
g.write_fastq("/tmp/c4_1.fq", r1); g.write_fastq("/tmp/c4_2.fq", r2)
PY
for dv in 0 0,0; do
  for t in 32 64; do
    /usr/bin/env bash -c "time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t $t --devices $dv -o /tmp/o_$dv.sam /tmp/c4ref.fa /tmp/c4_1.fq /tmp/c4_2.fq" 2> $O/cli_${dv}_t$t.err
    grep -E "real|index build|lane 0|lane 1" $O/cli_${dv}_t$t.err | cut -c1-400
  done
done
cmp /tmp/o_0.sam /tmp/o_0,0.sam && echo "lanes identical"
MM2REF_TIMING=1 $REPO/oracle/_ref/mm2ref -t 64 /tmp/c4ref.fa /tmp/c4_1.fq /tmp/c4_2.fq 2> $O/ref.err > /tmp/o_ref.sam; grep mm2ref $O/ref.err
cmp /tmp/o_0.sam /tmp/o_ref.sam && echo "reference identical"
