#!/bin/bash
# start-up cost of the drop-in on the C4 reference: index build breakdown (AL_TIMING), twice (cold / warm page cache)
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2idx}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
python3 - <<PY
import sys; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("c4"); g.write_fasta("/tmp/c4ref.fa", ref)
r1, r2 = g.simulate("c4", ref, ${PAIRS:-500000}, 4242)
g.write_fastq("/tmp/c4_1.fq", r1); g.write_fastq("/tmp/c4_2.fq", r2)
PY
for i in 1 2; do
  /usr/bin/env bash -c "time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t 32 -o /tmp/o.sam /tmp/c4ref.fa /tmp/c4_1.fq /tmp/c4_2.fq" 2> $O/cli_$i.err
  grep -E "real|index|lane 0" $O/cli_$i.err | cut -c1-300
done
/usr/bin/env bash -c "time AL_SERIAL_PARSE=1 AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t 32 -o /tmp/o2.sam /tmp/c4ref.fa /tmp/c4_1.fq /tmp/c4_2.fq" 2> $O/cli_serial.err
grep -E "real|index|lane 0" $O/cli_serial.err | cut -c1-300
cmp /tmp/o.sam /tmp/o2.sam && echo "identical"
