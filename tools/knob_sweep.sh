# usage (GPU box): bash tools/knob_sweep.sh  -> gpurun_out/knob_*.json   (scratch script for one-off knob sweeps of bench.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 2 3 4; do
  GASM_RULER_SHIFT=$v timeout -k 10 200 python bench.py --workload cfg1 --steps 200 --warmup 10 --no-cpu-baseline --breakdown > gpurun_out/knob_c1rs$v.json 2>> gpurun_out/knobs.err || exit 1
done
