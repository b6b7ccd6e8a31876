run() { env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=dict(r['other']); o[r['kernel']]={'avg_launch_ms':r['avg_launch_ms']}
print('$*', {k: v['avg_launch_ms'] for k,v in o.items()}, 'step', d['ms_per_step'])" || echo "$* failed"; }
run GASM_DBG_SCATTER=0
run GASM_DBG_SCATTER=2
run GASM_DBG_SCATTER=3
