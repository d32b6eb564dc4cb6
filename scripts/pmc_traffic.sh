#!/bin/bash
# HBM traffic of one step of the judged bench, per kernel: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE take
# 3 + 2 of the 4 TCC slots, MI355X_MICROARCH.md), counters in their own runs with --kernel-trace only.
# Usage: scripts/pmc_traffic.sh <tag>   -> gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}/*_counter_collection.csv
TAG=${1:-traffic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
