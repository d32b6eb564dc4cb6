#!/bin/bash
# Round 6, session 10: configs[3] (B = 16) per-kernel rocprofv3 summary -- where the GEMM family's 0.74 comes from.
TAG=${1:-r06t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -o trace -- python $ROOT/bench.py --config 3 --utterances 64 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/prof_c3_bench.json 2> $OUT/prof_c3.err; echo "rocprof c3 rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof_c3 $OUT/kernel_stats_c3.csv 12 > /dev/null 2>&1; head -60 $OUT/kernel_stats_c3.csv | cut -c1-200
find $OUT/prof_c3 -name "*kernel_trace.csv" -delete
