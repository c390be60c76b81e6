#!/bin/bash
# file-to-file rate of the stream driver against batch size / contexts / slots: CFG, PAIRS, RUNS="ctxs:slots:batch_reads ..." (batch 0 = the driver's own choice)
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r3sweep}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
CFG=${CFG:-c4}
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("$CFG"); g.write_fasta("/tmp/sref.fa", ref)
r1, r2 = g.simulate("$CFG", ref, ${PAIRS:-4000000}, 4242)
g.write_fastq("/tmp/s_1.fq", r1); g.write_fastq("/tmp/s_2.fq", r2)
PY
prev=""
for run in ${RUNS:-1:3:0}; do
  IFS=: read cx sl br <<< "$run"
  ev="AL_CTXS=$cx AL_SLOTS=$sl"; [ "$br" != "0" ] && ev="$ev AL_BATCH_READS=$br"
  /usr/bin/env bash -c "time AL_TIMING=1 $ev ${ENVX:-} $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_$run.sam /tmp/sref.fa /tmp/s_1.fq /tmp/s_2.fq" 2> $O/${CFG}_$run.err
  echo "== $CFG ctxs:slots:batch = $run"; grep -E "^real|index build|stream|context" $O/${CFG}_$run.err | cut -c1-700
  [ -n "$prev" ] && { cmp /tmp/so_$run.sam $prev && echo "identical to the previous run"; rm -f $prev; }
  prev=/tmp/so_$run.sam
done
