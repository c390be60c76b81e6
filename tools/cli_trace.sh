#!/bin/bash
# kernel timeline of the drop-in process (stream driver) under rocprofv3: which kernels, how busy the GPU is between the first and the last
# usage: CFG=c4 PAIRS=2000000 ENVX="AL_CTXS=1 AL_BATCH_READS=1048576" tools/cli_trace.sh <tag>
TAG=${1:-r3trace}
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/$TAG; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
CFG=${CFG:-c4}
python3 - <<PY
import sys, time; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("$CFG"); g.write_fasta("/tmp/sref.fa", ref)
r1, r2 = g.simulate("$CFG", ref, ${PAIRS:-2000000}, 4242)
g.write_fastq("/tmp/s_1.fq", r1); g.write_fastq("/tmp/s_2.fq", r2)
PY
cd /tmp && export TMPDIR=/tmp
export AL_TIMING=1 AL_NO_FAST_EXIT=1 ${ENVX:-}
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_trace.sam /tmp/sref.fa /tmp/s_1.fq /tmp/s_2.fq 2> $O/trace.err
grep -E "stream|context" $O/trace.err | cut -c1-600
python3 $REPO/tools/trace_busy.py $O/trace > $O/busy.md; cat $O/busy.md | head -60
find $O -name "*.csv" -size +40M -delete
