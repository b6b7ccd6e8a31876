#!/bin/bash
# usage: tools/variant_bench.sh <log> <variant> ...   ("-" = the tree's own libgasm.so); one bench line summary per variant
log=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset GASM_LIBGASM; else export GASM_LIBGASM=$PWD/tools/micro/libgasm_$v.so; fi
  echo "== variant [$v]" >> $log
  timeout -k 10 150 python bench.py --no-cpu-baseline --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j[\"roofline\"]; print(j[\"ms_per_step\"], r[\"kernel\"], r[\"avg_launch_ms\"], {k:v[\"avg_launch_ms\"] for k,v in r[\"other\"].items()})" >> $log
done
cat $log
