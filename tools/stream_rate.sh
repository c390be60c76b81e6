#!/bin/bash
# file-to-file rate of the drop-in (FASTQ files on /tmp -> SAM file): stream driver (parse / pack / SAM text on the GPU) against the
# host driver, slots swept.  CFG=c4|c2|c3, PAIRS, SLOTS="2 3", TAG
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r3stream}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
CFG=${CFG:-c4}
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("$CFG"); g.write_fasta("/tmp/sref.fa", ref)
r1, r2 = g.simulate("$CFG", ref, ${PAIRS:-4000000}, 4242)
g.write_fastq("/tmp/s_1.fq", r1); g.write_fastq("/tmp/s_2.fq", r2)
PY
ls -la /tmp/s_1.fq /tmp/s_2.fq | awk '{print $5}'
for sl in ${SLOTS:-2 3}; do
  /usr/bin/env bash -c "time AL_TIMING=1 AL_SLOTS=$sl ${ENVX:-} $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_$sl.sam /tmp/sref.fa /tmp/s_1.fq /tmp/s_2.fq" 2> $O/stream_${CFG}_s$sl.err
  echo "== slots $sl"; grep -E "^real|index build|stream|lane" $O/stream_${CFG}_s$sl.err | cut -c1-600
done
if [ -z "$NOHOST" ]; then
  /usr/bin/env bash -c "time AL_TIMING=1 AL_HOST_IO=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_host.sam /tmp/sref.fa /tmp/s_1.fq /tmp/s_2.fq" 2> $O/host_${CFG}.err
  echo "== host driver"; grep -E "^real|index build|lane" $O/host_${CFG}.err | cut -c1-400
  for sl in ${SLOTS:-2 3}; do cmp /tmp/so_$sl.sam /tmp/so_host.sam && echo "slots $sl: identical to the host driver"; done
fi
ls -la /tmp/so_*.sam | awk '{print $5, $9}'
