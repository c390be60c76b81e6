#!/bin/bash
# kernel trace of one bench configuration: top kernels by total time and the longest single launches.  usage: trace_top.sh <config> [bench args]
CFG=${1:-c5}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}; O=$REPO/gpurun_out/trace_$CFG; mkdir -p $O
export AL_REF_CACHE=/tmp/alcache
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 $REPO/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --f2f-pairs 0 "$@" > $O/bench.json 2> $O/err.txt
python3 - $O <<'PY'
import sys,glob,csv,collections
rows=[]
for fn in glob.glob(sys.argv[1]+'/t/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Stream_Id','')))
rows.sort()
# the last step = the last occurrence of k_sketch onwards
starts=[i for i,r in enumerate(rows) if r[2].startswith('k_sketch') or 'k_sketch' in r[2][:40]]
i0=starts[-1] if starts else 0
rows=rows[i0:]
t0=rows[0][0]
tot=collections.defaultdict(lambda:[0,0,0])
for s,e,k,st in rows:
    t=tot[k[:70]]; t[0]+=1; t[1]+=e-s; t[2]=max(t[2],e-s)
print("span %.1f ms, %d launches"%((max(r[1] for r in rows)-t0)/1e6,len(rows)))
for k,v in sorted(tot.items(), key=lambda x:-x[1][1])[:28]: print("%8.2f ms total %5d launches, longest %8.2f ms  %s"%(v[1]/1e6,v[0],v[2]/1e6,k))
print("longest launches:")
for s,e,k,st in sorted(rows,key=lambda r:r[0]-r[1])[:25]: print("  +%8.2f ms  %8.2f ms  stream %s  %s"%((s-t0)/1e6,(e-s)/1e6,st,k[:80]))
PY
find $O -name "*.csv" -size +6M -delete
