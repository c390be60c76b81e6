#!/bin/bash
# timing experiments of the lane chaining kernels: per-kernel stats under AL_DBG2 = 0, 1, 2, 4, 7 (results invalid for != 0)
export AL_REF_CACHE=/tmp/alcache
for v in ${VALS:-0 1 2 4 7}; do
  AL_DBG2=$v timeout ${TMO:-150} bash tools/ktrace.sh dbg2_$v --f2f-pairs 0 --steps 2 --warmup 1 ${BARGS:-} > gpurun_out/dbg2_$v.txt 2>&1
  echo "== AL_DBG2=$v"; grep -E "k_chain_lds|k_chain_tile6|k_u_compact" gpurun_out/dbg2_$v.txt | cut -c1-130
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/kt_dbg2_$v/bench.json").read().strip().splitlines()[-1])
    print("  ms/step", round(d["ms_per_step"],2), {k:round(x,2) for k,x in d["stages_ms"].items() if k in ("anchor_sort_blk","chain_tile","chain_deferred","rechain","chain_ties")})
except Exception as e: print("  bench line failed", e)
PY
done
