#!/bin/bash
# round-2 GPU session 1: reduced-precision kernels -- parity tests, microbench sweep, bench lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2a; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "reduced_precision" -p no:cacheprovider > $OUT/pytest_lp_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $OUT/pytest_lp_kernels.log
timeout 1200 python -m pytest tests/test_gpu_precision.py tests/test_gpu_engine.py -m gpu -q -s -k "precision or configs or whisper_15s or two_windows" -p no:cacheprovider > $OUT/pytest_precision.log 2>&1; echo "precision rc=$?"; grep -E "err|passed|failed|Error|assert" $OUT/pytest_precision.log | tail -30
timeout 600 python scripts/microbench.py lp > $OUT/microbench_lp.log 2>&1; echo "microbench rc=$?"; cat $OUT/microbench_lp.log | tail -150
for P in f32 bf16x3 bf16 f16; do
  timeout 600 python bench.py --steps 20 --warmup 3 --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; echo "bench $P rc=$?"; cat $OUT/bench_$P.json | cut -c1-1500
done
