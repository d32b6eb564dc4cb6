#!/bin/bash
# round-2 GPU session 3: narrow-stage MFMA kernel + streaming decoder
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2d; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "snake_conv" -p no:cacheprovider > $OUT/pytest_snake.log 2>&1; echo "snake rc=$?"; tail -3 $OUT/pytest_snake.log
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -s -k "streaming or full_10s or generator_base or batch16" -p no:cacheprovider > $OUT/pytest_engine.log 2>&1; echo "engine rc=$?"; grep -E "passed|failed|rror|err|tiled" $OUT/pytest_engine.log | tail -8
timeout 600 python scripts/microbench.py ampgroup > $OUT/microbench_ampgroup.log 2>&1; cat $OUT/microbench_ampgroup.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-2200 $OUT/bench.json
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-2600 $OUT/bench_c2.json
