#!/bin/bash
# instruction counters of the DP kernels, packed form and byte-packed form (AL_DBG bit 17)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_dp; mkdir -p $OUT
export AL_REF_CACHE=/tmp/alcache
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 1 --warmup 0 --f2f-pairs 0 --pairs 250000"
for mode in pk old; do
  if [ $mode = old ]; then export AL_DBG=$((1<<17)); else unset AL_DBG; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d $OUT/$mode --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/$mode.json 2> $OUT/$mode.err
  python3 - $OUT/$mode <<'PY'
import sys,glob,csv,collections
d=sys.argv[1]
f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]
        if 'ext_dp' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); acc[k]['n_'+r['Counter_Name']]+=1
for k,v in acc.items(): print(k, {a:(b if a.startswith('n_') else '%.3e'%b) for a,b in v.items()})
PY
  find $OUT/$mode -name "*.csv" -size +6M -delete
done
