#!/bin/bash
# round-2 GPU session 14: full validation after chunk streams / window batching
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2x; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json
timeout 600 python scripts/wav2wav_times.py 10 > $OUT/wav2wav.log 2>&1; tail -4 $OUT/wav2wav.log
