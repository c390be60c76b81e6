#!/bin/bash
# one traced bench run on C4 (no CPU legs): stage times, kernel stats, AL_TRACE diagnostics
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/${TAG:-r2q}; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd /tmp && export TMPDIR=/tmp
AL_TRACE=1 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 --pairs ${PAIRS:-500000} ${BENCH_ARGS:-} > $O/c4_trace.json 2> $O/c4_trace.err
find $O -name "*.csv" -size +4M -delete
grep "trace: regs" $O/c4_trace.err | tail -2
python3 - <<PY
import json,csv,glob
d=json.load(open("$O/c4_trace.json"))
print(d["value"], d["ms_per_step"])
print({k:round(v,2) for k,v in d["stages_ms"].items() if v>0.3})
print({k:v for k,v in d["counters"].items() if k in("heap_fallback","chain_fallback","side_stream_ms")})
f=glob.glob("$O/trace/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:18]:
    print(r['Name'][:60], r['Calls'], int(r['TotalDurationNs'])//1000, int(float(r['AverageNs']))//1000, r['Percentage'])
PY
