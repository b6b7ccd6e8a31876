# Ablations / tuning sweeps of the two dominant kernels on the GPU box (from the repo root): bash tools/ablate.sh
run() { env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=dict(r['other']); o[r['kernel']]={'avg_launch_ms':r['avg_launch_ms']}
print('$*', {k: v['avg_launch_ms'] for k,v in o.items()}, 'step', d['ms_per_step'])" || echo "$* failed"; }
run GASM_SCATTER_WGS=8
run GASM_SCATTER_WGS=2
run GASM_SCATTER_WGS=32
run GASM_DBG_PADM=7          # runs padded to 64 bytes instead of whole lines
run GASM_DBG_PADM=0          # no padding
run GASM_DEDUP_TBL=4096
run GASM_DBG_DEDUP=1         # loads + hashing only (wrong results)
run GASM_DBG_DEDUP=2         # no ordering / write-back (wrong results)
run GASM_RANK_GLOBAL=1       # list ranking without the LDS kernel
