#!/bin/bash
# Round 6, session 7: the eight-wave 128x80 tile: hardware tests (bit-identical to the four-wave tile), per-launch times on the Whisper shapes.
TAG=${1:-r06n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or gemm" 2>&1 | tail -n 3 | tee $OUT/pytest_kernels.log
timeout 600 python scripts/microbench.py w8 2>&1 | grep -v amdgpu.ids | tee $OUT/w8.log
echo "== done"
