#!/bin/bash
# Round 6, session 3: cycles per K-step, prologue / epilogue cycles and the shader clock of the Whisper MLP-up (n_out 5120) and QKV (3840)
# launches INSIDE the captured pipeline: one clip at a time and 4 clips in flight.
TAG=${1:-r06e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for v in kt5120 kt3840; do
  K=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$v.so
  echo "== $v, 4 clips in flight"
  SVCMI_KTRACE_CLOCK=1 SVCMI_LIB=$K timeout 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/${v}_inflight.json 2> $OUT/${v}_inflight.err; grep "ktrace sample\|ms_per_step" $OUT/${v}_inflight.err
  echo "== $v, one clip at a time"
  SVCMI_KTRACE_CLOCK=1 SVCMI_LIB=$K timeout 200 python bench.py --inflight 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/${v}_single.json 2> $OUT/${v}_single.err; grep "ktrace sample\|ms_per_step" $OUT/${v}_single.err
done
echo "== done"
