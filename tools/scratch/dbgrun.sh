for d in ${DBGS:-0}; do echo "AL_DBG=$d"; AL_DBG=$d timeout 300 python bench.py --pairs ${PAIRS:-500000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stages_ms']; print('  total %.1f ms  ' % d['ms_per_step'] + ' '.join('%s %.1f' % (k, v) for k, v in s.items() if v > 0.2))"; done
