#!/bin/bash
# usage: exp.sh tag "ENV=.. ENV2=.." cfg pairs
tag=$1; envs=$2; cfg=$3; pairs=$4
mkdir -p gpurun_out/$tag
env $envs python bench.py --steps 4 --warmup 2 --no-cpu-baseline --f2f-pairs 0 --config $cfg --pairs $pairs 2>gpurun_out/$tag/err.txt | tail -1 > gpurun_out/$tag/out.json
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/out.json"))
print("$tag [$envs] $cfg", round(d["ms_per_step"],2), {k:round(v,1) for k,v in d["stages_ms"].items() if k in ("chain_ties","rechain","chain_tile","anchor_sort_blk","regs")})
PY
