#!/bin/bash
# One GPU-box session: smoke, gpu tests, bench, rocprof kernel trace.  Usage: scripts/gpu_check.sh [tag]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6 | tee $OUT/rocminfo.txt
nproc | tee $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" | tee -a $OUT/nproc.txt
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
if [ "$SKIP_TESTS" != "1" ]; then
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
fi
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-20} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
if [ "$SKIP_PROF" != "1" ]; then
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $ROOT
find $OUT/prof -name "*stats*" | head; 
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -25 "$F"
# keep the merge-back small: drop the raw per-dispatch trace if huge
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
fi
echo "== done"
