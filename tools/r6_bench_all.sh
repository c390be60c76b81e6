#!/bin/bash
# the round's bench lines: tools/r6_bench_all.sh <tag>  ->  gpurun_out/<tag>_bench_{c2,c3,c5,c4_default}.json (C2 / C3 / C5 with their CPU comparator, no file-to-file leg; C4 = the default command)
tag=${1:-r06b}
export AL_REF_CACHE=/tmp/alcache TMPDIR=/tmp
for c in c2 c3 c5; do
  p=1000000; [ $c = c5 ] && p=500000
  timeout 500 python bench.py --config $c --pairs $p --steps 10 --warmup 3 --f2f-pairs 0 > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err; tail -c 300 gpurun_out/${tag}_bench_$c.err
done
timeout 900 python bench.py > gpurun_out/${tag}_bench_c4_default.json 2> gpurun_out/${tag}_bench_c4.err; tail -c 300 gpurun_out/${tag}_bench_c4.err
python - <<PY
import json
for c in ("c2","c3","c5","c4_default"):
    try:
        d=json.loads(open("gpurun_out/${tag}_bench_%s.json"%c).read().strip().splitlines()[-1]); print(c, round(d["ms_per_step"],2), round(d["value"]), d.get("parity_sample"), (d.get("file_to_file") or {}).get("reads_per_s"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(c, "failed", e)
PY
