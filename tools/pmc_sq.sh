# usage (GPU box): bash tools/pmc_sq.sh <tag> [bench args]  -> gpurun_out/pmc_sq_<tag>.txt  (SQ counters of the build kernels, one pass)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/pmc_sq_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_sq_$tag.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_sq_$tag k_bucket k_tile > $out/pmc_sq_$tag.txt 2>&1
rm -rf $out/pmc_sq_$tag
