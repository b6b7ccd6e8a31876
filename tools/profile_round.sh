# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/: bench line (+cpu baseline, per-kernel table), rocprofv3 kernel stats of cfg2 / cfg4 (128-bit
#    keys) / the pooled mode on one GPU, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, summarised per kernel), one SQ
#    pass (VALU / LDS activity, LDS bank conflicts, waves).  Raw counter CSVs are summarised and deleted (size).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
head=$(cat .git_head 2>/dev/null || echo unknown)
timeout -k 10 400 python bench.py --breakdown > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --breakdown > $out/bench_cfg4.json 2> $out/bench_cfg4.err || exit 1
timeout -k 10 300 python bench.py --workload cfg1 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_cfg1.json 2> $out/bench_cfg1.err || exit 1
timeout -k 10 300 python bench.py --mode pooled --steps 5 --warmup 1 --no-cpu-baseline --breakdown > $out/bench_pooled_n1.json 2> $out/bench_pooled_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/stats_$name.log 2>&1 || return 2
  f=$(find $out/stats_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${name}_kernel_stats.csv && rm -rf $out/stats_$name
}
stats cfg2 || exit 2      # (the default flags: the same command as the bench line — durations are residence times: steps overlap)
# one step in flight and every kernel alone on the chip: the kernels' own durations (the bench line's roofline.one_step_in_flight)
export GASM_PINGPONG=0 GASM_SCORE_LANE=0
stats cfg2_one_step_in_flight --alone-steps 0 || exit 2
stats cfg4 --workload cfg4 --steps 5 --warmup 1 || exit 2
stats pooled_n1 --mode pooled --steps 3 --warmup 1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 4
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out "$head" > $out/pmc_traffic.json   # (counter passes: one step in flight, see the export above)
rm -rf $out/pmc_fetch $out/pmc_write
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_sq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_sq k_bucket k_tile k_rank k_score > $out/pmc_sq_summary.txt 2>&1
rm -rf $out/pmc_sq
unset GASM_PINGPONG GASM_SCORE_LANE
echo "profile $tag done"
