#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2e}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd $REPO
timeout 1200 python3 -m pytest tests/test_gpu_sam.py -m gpu -x -q -k "multi_lane or cli_sam or large_fragment" > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
# two lanes on one GPU: does overlapping transfers / host work with kernels pay?  (C2 files, 2 M pairs)
python3 - <<PY
import sys; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("c2"); g.write_fasta("/tmp/c2ref.fa", ref)
r1, r2 = g.simulate("c2", ref, 2000000, 77)
g.write_fastq("/tmp/c2_1.fq", r1); g.write_fastq("/tmp/c2_2.fq", r2)
PY
for dv in 0 0,0 0,0,0; do
  for t in 16 48; do
    /usr/bin/env bash -c "time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t $t --devices $dv -o /tmp/o_$dv.sam /tmp/c2ref.fa /tmp/c2_1.fq /tmp/c2_2.fq" 2> $O/lanes_${dv}_t$t.err
    grep -E "real|lane 0" $O/lanes_${dv}_t$t.err | head -3
  done
done
cmp /tmp/o_0.sam /tmp/o_0,0.sam && cmp /tmp/o_0.sam /tmp/o_0,0,0.sam && echo "lanes identical"
