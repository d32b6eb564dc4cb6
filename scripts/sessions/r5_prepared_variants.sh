#!/bin/bash
# First GPU session of round 5: the three build experiments round 4 prepared (all default-off).
# Build the variants HERE first (they travel with the snapshot):
#   scripts/build_variant.sh a1   -DSVCMI_TILE_ADDR32=1
#   scripts/build_variant.sh a2   -DSVCMI_TILE_ADDR32=2
#   scripts/build_variant.sh il   -DSVCMI_SNAKE_INTERLEAVE=1
#   scripts/build_variant.sh a2il -DSVCMI_TILE_ADDR32=2 -DSVCMI_SNAKE_INTERLEAVE=1
# then: scripts/gpu.sh --timeout 1500 -- 'bash scripts/sessions/r5_prepared_variants.sh r05a'
# Per variant: the half-step / SnakeAlias kernel tests (bit-identity and reference checks at full lengths), then the microbench.
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for V in default a1 a2 il a2il; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so; [ -f $SVCMI_LIB ] || { echo "$V: not built"; continue; }; fi
  timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake or amp_block" > $OUT/pytest_$V.log 2>&1; echo "$V pytest rc=$?"; tail -1 $OUT/pytest_$V.log
  timeout 300 python scripts/microbench.py amplp snake > $OUT/micro_$V.log 2>&1; echo "$V microbench rc=$?"
  grep -E "amplp .* B=1 d=1 amp_u= 1|amplp .* B=4 d=1 amp_u= 1|^snake C=(40|20|10) " $OUT/micro_$V.log | sed 's/  max diff.*//'
done
