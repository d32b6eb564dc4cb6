#!/bin/bash
# MFMA / LDS / wave-state counters of one eager step of the judged bench, per kernel (counters in their own runs,
# --kernel-trace only; BENCH_ARGS="--precision bf16x3" profiles a reduced-precision step).  Usage: scripts/pmc_bench.sh <tag> -> gpurun_out/<tag>/<pass>/*_counter_collection.csv
TAG=${1:-pmcbench}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
    python $ROOT/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/pass$i.log 2>&1
  echo "pass$i rc=$?"
done
