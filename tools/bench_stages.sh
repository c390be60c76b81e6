#!/bin/bash
# resident bench line + stage table (no CPU baseline, no file-to-file leg): tools/bench_stages.sh <tag> [bench args]
tag=$1; shift
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --f2f-pairs 0 "$@" 2>gpurun_out/$tag.err | tail -1 > gpurun_out/$tag.json
python - <<PY
import json
d=json.load(open("gpurun_out/$tag.json"))
print(d["ms_per_step"], d["value"])
for k,v in d.get("stages_ms",{}).items(): print(" ",k,round(v,2))
PY
