#!/bin/bash
# round-3 profile collection for the judged line (configs[1], fp32): rocprofv3 kernel stats of the single-stream run, HBM traffic (two PMC
# passes) and MFMA / LDS / wave-state counters (two PMC passes), each summarised into gpurun_out/<tag>/ for profiles/.
TAG=${1:-r03g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof1_bench.json 2> $OUT/prof1.err; echo "rocprof single rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof1 $OUT/kernel_stats.csv 11 > /dev/null 2>&1; head -24 $OUT/kernel_stats.csv
find $OUT/prof1 -name "*kernel_trace.csv" -delete
bash scripts/pmc_traffic.sh $TAG/traffic
python scripts/traffic_summary.py $OUT/traffic $OUT/traffic.json 3 2>&1 | tail -5
bash scripts/pmc_bench.sh $TAG/pmc
python scripts/pmc_summary.py $OUT/pmc $OUT/pmc.json 2>&1 | tail -5
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
