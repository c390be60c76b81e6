#!/bin/bash
# steady-state rate of the drop-in where the GPU is not the limit (C2, yeast-sized): PAIRS pairs from files, host threads swept
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2host}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("${CFG:-c2}"); g.write_fasta("/tmp/href.fa", ref)
r1, r2 = g.simulate("${CFG:-c2}", ref, ${PAIRS:-4000000}, 4242)
g.write_fastq("/tmp/h_1.fq", r1); g.write_fastq("/tmp/h_2.fq", r2)
PY
for t in ${TS:-16 32 64}; do
  for dv in ${DVS:-0 0,0}; do
    /usr/bin/env bash -c "time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t $t --devices $dv ${KARG:-} -o /tmp/ho_$dv.sam /tmp/href.fa /tmp/h_1.fq /tmp/h_2.fq" 2> $O/cli_${dv}_t$t.err
    echo "-t $t --devices $dv: $(grep -E '^real' $O/cli_${dv}_t$t.err)"; grep -E "lane 0" $O/cli_${dv}_t$t.err | cut -c1-330
  done
done
ls -la /tmp/ho_0.sam | awk '{print $5}'
