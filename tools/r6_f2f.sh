#!/bin/bash
# file-to-file leg under several environments: tools/r6_f2f.sh "<env set 1>;<env set 2>;..."   (C4 reference + PAIRS pairs on /dev/shm -> airlift-align -> SAM in /tmp)
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-f2f6}; mkdir -p $O
CFG=${CFG:-c4}
python3 - <<PY
import sys; sys.path.insert(0, "$REPO/tools")
import gen_synth as g
ref = g.build_reference("$CFG"); g.write_fasta("/dev/shm/sref.fa", ref)
r1, r2 = g.simulate("$CFG", ref, ${PAIRS:-6250000}, 4242)
g.write_fastq("/dev/shm/s_1.fq", r1); g.write_fastq("/dev/shm/s_2.fq", r2)
PY
IFS=';' read -ra SETS <<< "${1:-X=1}"
i=0
for envx in "${SETS[@]}"; do
  i=$((i+1)); sleep ${SLEEP:-4}
  ( time timeout 120 env AL_TIMING=1 $envx $REPO/airlift_amd/bin/airlift-align -ax sr -t ${T:-32} -o /tmp/so_$i.sam /dev/shm/sref.fa /dev/shm/s_1.fq /dev/shm/s_2.fq ) 2> $O/run_$i.err
  echo "== run $i ($envx)"; grep -E "^real|stream pipeline:|reserve|-> batches|halv" $O/run_$i.err | cut -c1-420
  [ $i -gt 1 ] && (cmp /tmp/so_1.sam /tmp/so_$i.sam && echo "   identical to run 1")
  [ $i -gt 1 ] && rm -f /tmp/so_$i.sam
done
