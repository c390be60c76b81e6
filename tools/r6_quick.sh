#!/bin/bash
# round-6 iteration call: tools/r6_quick.sh <tag> [pytest args...]; a parity subset, then resident lines of the configurations in R6_CFGS (default c4)
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp AL_REF_CACHE=/tmp/alcache
if [ $# -gt 0 ]; then ( timeout ${R6_TEST_TIMEOUT:-1500} python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log; fi
for cfg in ${R6_CFGS:-c4}; do
  pairs=1000000; [ $cfg = c5 ] && pairs=500000
  timeout 600 python bench.py --steps ${R6_STEPS:-5} --warmup 2 --no-cpu-baseline --f2f-pairs 0 --config $cfg --pairs $pairs 2>gpurun_out/$tag/$cfg.err | tail -1 > gpurun_out/$tag/$cfg.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/$cfg.json"))
    print("$cfg", round(d["ms_per_step"],2), round(d["value"]), "parity", d.get("parity_sample"))
    print("  ", {k:round(v,2) for k,v in d.get("stages_ms",{}).items() if v>0.3})
except Exception as e: print("$cfg failed", e)
PY
done
