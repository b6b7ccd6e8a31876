# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/: bench line (+cpu baseline, per-kernel table), rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE PMC passes
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --breakdown > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 4
echo "profile $tag done"
