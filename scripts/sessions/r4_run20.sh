#!/bin/bash
# round-4 GPU session 20: profile of the configs[2] step under the mixed policy at the final HEAD: rocprofv3 kernel stats, PMC HBM traffic,
# MFMA / wave-state / LDS counters (every run also executes ONE fp32 step: bench.py's live precision check)
TAG=${1:-r04w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="--config 2 --eager --inflight 1 --no-roofline --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py $ARGS --steps 3 --warmup 1 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $ROOT; python scripts/prof_summary.py $OUT/prof $OUT/kernel_stats_c2_mixed.csv 5 > /dev/null 2>&1; head -24 $OUT/kernel_stats_c2_mixed.csv; find $OUT/prof -name "*kernel_trace.csv" -delete
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/traffic/$C -o pmc -- python $ROOT/bench.py $ARGS --steps 2 --warmup 1 > $OUT/traffic_$C.log 2>&1; echo "$C rc=$?"
done
cd $ROOT; python scripts/traffic_summary.py $OUT/traffic $OUT/traffic_c2_mixed.json 5 2>&1 | tail -4
BENCH_ARGS="--config 2 --inflight 1" bash scripts/pmc_bench.sh $TAG/pmc
python scripts/pmc_summary.py $OUT/pmc $OUT/pmc_c2_mixed.json 2>&1 | tail -8
find $OUT -name "*counter_collection.csv" -delete
