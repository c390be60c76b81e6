for d in ${DBGS:-0 1 128}; do echo "AL_DBG=$d"; AL_DBG=$d timeout 300 python bench.py --pairs ${PAIRS:-500000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stages_ms']; print('  chain %.1f align %.1f total %.1f  parity n/a' % (s['chain'], s['align'], d['ms_per_step']))"; done
