# usage: tools/pmc.sh <tag> <counter> [counter...]   -> gpurun_out/pmc_<tag>/ (one rocprofv3 --pmc pass of a short bench run)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
