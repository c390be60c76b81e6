#!/bin/bash
# C2 / C3 bench lines + a kernel timeline of C3: tools/r5_small.sh <tag>
tag=$1; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
for cfg in c2 c3; do
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --f2f-pairs 0 --config $cfg 2>gpurun_out/$tag/$cfg.err | tail -1 > gpurun_out/$tag/$cfg.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/$cfg.json"))
    print("$cfg", round(d["ms_per_step"],2), round(d["value"]), d["config"].get("pairs"))
    print("  ", {k:round(v,2) for k,v in d.get("stages_ms",{}).items() if v>0.2})
except Exception as e: print("$cfg failed", e)
PY
done
TLARGS="--config c3" tools/tl.sh ${tag}_c3 100
