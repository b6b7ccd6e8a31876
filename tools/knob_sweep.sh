# usage (GPU box): bash tools/knob_sweep.sh  -> gpurun_out/knob_*.json   (scratch script for one-off knob sweeps of bench.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 2 3; do
  GASM_DBG_BBITS_ADD=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown > gpurun_out/knob_bb$v.json 2>> gpurun_out/knobs.err || exit 1
  echo "$v done" >> gpurun_out/knobs.log
done
GASM_DEDUP_TBL=4096 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown > gpurun_out/knob_t4096.json 2>> gpurun_out/knobs.err || exit 1
