#!/bin/bash
# Round 6, session 15: per-kernel profile of the reduced-precision configs[1] lines, one clip at a time (rocprofv3 serialises the lanes anyway)
TAG=${1:-r06zb}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for P in mixed f16; do
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -o trace -- python $ROOT/bench.py --precision $P --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_${P}_bench.json 2> $OUT/prof_$P.err; echo "rocprof $P rc=$?"
cd $ROOT; python scripts/prof_summary.py $OUT/prof_$P $OUT/kernel_stats_$P.csv 11 > /dev/null 2>&1; head -34 $OUT/kernel_stats_$P.csv | cut -c1-150
find $OUT/prof_$P -name "*kernel_trace.csv" -delete
done
