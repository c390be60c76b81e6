#!/bin/bash
# end-to-end wall clock of the CLI (FASTQ in -> SAM out) against the CPU comparator on the same files: tools/e2e.sh [pairs]
PAIRS=${1:-500000}
export AL_PG_PLAIN=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}; D=/tmp/al_e2e; mkdir -p $D
python3 - <<PY
import sys; sys.path.insert(0, "$REPO/tools")
import gen_synth as g, numpy as np
ref = g.build_reference("${CONFIG:-c2}"); g.write_fasta("$D/ref.fa", ref)
r1, r2 = g.simulate("${CONFIG:-c2}", ref, $PAIRS, 77, read_len=${RLEN:-150})
g.write_fastq("$D/r_1.fq", r1, "realigned_"); g.write_fastq("$D/r_2.fq", r2, "realigned_")
PY
cd $D
TIMEFORMAT="airlift-align --version (process + library load): %R s"; time $REPO/airlift_amd/bin/airlift-align --version > /dev/null
for t in ${T1:-1} ${THREADS:-16}; do
  TIMEFORMAT="airlift-align -t $t: %R s wall, %U s user ($PAIRS pairs)"; time AL_TIMING=1 $REPO/airlift_amd/bin/airlift-align -ax sr -t $t ref.fa r_1.fq r_2.fq > out_gpu.sam 2> err_gpu.txt; grep airlift err_gpu.txt
done
if [ -x $REPO/oracle/_ref/mm2ref ]; then
  TIMEFORMAT="mm2ref -t $(nproc): %R s wall, %U s user"; time $REPO/oracle/_ref/mm2ref -t $(nproc) ref.fa r_1.fq r_2.fq > out_cpu.sam 2> err_cpu.txt
  cmp out_gpu.sam out_cpu.sam && echo "SAM identical ($(wc -l < out_gpu.sam) lines)"
fi
# BAM modes of the drop-in on the same files (no samtools in this image to time the pipeline it replaces)
for m in --bam --sorted-bam; do
  TIMEFORMAT="airlift-align $m -l 5 -t ${THREADS:-16}: %R s wall"; time $REPO/airlift_amd/bin/airlift-align -ax sr $m -l 5 -t ${THREADS:-16} ref.fa r_1.fq r_2.fq > out.bam 2>/dev/null; ls -la out.bam | awk '{print "  " $5 " bytes"}'
done
