#!/bin/bash
# CREPE with its short layers as dense GEMMs over the frames + the long-K tile rule: tests of the extractor, drop-in timings in both F0 precisions
TAG=${1:-r03v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "crepe or extractors or wav_to_wav or cli or reduced_precision" > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_subset.log
timeout 600 python scripts/dropin_times.py 10 bf16x3 > $OUT/dropin_times.log 2>&1; tail -12 $OUT/dropin_times.log
timeout 600 python scripts/dropin_times.py 10 f16 > $OUT/dropin_times_f0_f16.log 2>&1; tail -7 $OUT/dropin_times_f0_f16.log
