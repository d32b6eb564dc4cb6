#!/bin/bash
# Round 6, session 13: soak of a candidate library (SVCMI_LIB) against the in-flight corruption of snake_alias_kernel beside the fp16 fused half-step:
# the isolated pair (30 replays x 40 launches), and whole conversions through ClipLanes, two in flight, in the precision mixes that differed in r06y.
TAG=${1:-r06y}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
L=$OUT/soak_${2:-default}.log; : > $L
F32="enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f32,amp4=f32"
PROBE_SET=first PROBE_DUMP=0 PROBE_REPS=30 timeout 300 python scripts/lp_concurrency_probe10.py 2>&1 | grep probe10 >> $L
for P in "f16 f32" "bf16x3 f32" "mixed:enc=f32,flow=f32,ups=f16,amp0=f16,amp1=f16,amp2=f16,amp3=f32,amp4=f32 f32" "mixed:enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f16,amp4=f16 f32" "mixed f16" "f16 f16"; do
  for rep in 1 2; do timeout 300 python scripts/clip_lanes_matrix.py $P 2>&1 | grep matrix >> $L; done
done
cat $L
