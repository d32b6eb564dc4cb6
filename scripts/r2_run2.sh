#!/bin/bash
# round-2 GPU session 2: whole gpu suite, judged bench + rocprof, the other BASELINE configs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2b; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" $OUT/pytest_gpu.log | tail -8; grep -E "stress|config0|pred_ppg 2|\{'logmel" $OUT/pytest_gpu.log | tail
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
for C in 2 3 4; do
  timeout 900 python bench.py --config $C > $OUT/bench_c$C.json 2> $OUT/bench_c$C.err; echo "bench config $C rc=$?"; cut -c1-1800 $OUT/bench_c$C.json; tail -2 $OUT/bench_c$C.err
done
timeout 600 python bench.py --config 2 --precision bf16x3 --no-roofline > $OUT/bench_c2_bf16x3.json 2>> $OUT/bench_c2.err; cut -c1-700 $OUT/bench_c2_bf16x3.json
timeout 600 python bench.py --config 2 --precision f32 --no-roofline > $OUT/bench_c2_f32.json 2>> $OUT/bench_c2.err; cut -c1-700 $OUT/bench_c2_f32.json
timeout 600 python bench.py --config 4 --precision f32 --no-roofline > $OUT/bench_c4_f32.json 2>> $OUT/bench_c4.err; cut -c1-700 $OUT/bench_c4_f32.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof $OUT/kernel_stats.csv 12 > /dev/null 2>&1; head -30 $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
