#!/bin/bash
# round-5 iteration call: tools/r5_quick.sh <tag> [tests...]; logs under gpurun_out/<tag>/
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
( timeout 900 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/$tag/pytest.log
for cfg in ${R5_CFGS:-c5 c4}; do
  pairs=1000000; [ $cfg = c5 ] && pairs=500000
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --f2f-pairs 0 --config $cfg --pairs $pairs 2>gpurun_out/$tag/$cfg.err | tail -1 > gpurun_out/$tag/$cfg.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/$cfg.json"))
    print("$cfg", round(d["ms_per_step"],2), round(d["value"]))
    print("  ", {k:round(v,2) for k,v in d.get("stages_ms",{}).items() if v>0.3})
    print("  ", d.get("counters"))
except Exception as e: print("$cfg failed", e)
PY
done
cat gpurun_out/$tag/pytest.log
