#!/bin/bash
# Round 5, GPU session 14: the mid-barrier K loop of the fp32 GEMM (-DSVCMI_GEMM_MIDBAR=1: the barrier that publishes tile it+1 before the last
# sub-step of tile it, next tile's first fragments requested under the last sub-step's MFMAs) against the default loop
TAG=${1:-r05o}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
SVCMI_LIB=$EXP/libsvcmi_midbar.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "conv_gemm or grouped or two_deep or splitk" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -1 $OUT/pytest_kernels.log
SVCMI_LIB=$EXP/libsvcmi_midbar.so timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -p no:cacheprovider > $OUT/pytest_engine.log 2>&1; echo "engine rc=$?"; tail -1 $OUT/pytest_engine.log
for V in default midbar; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$EXP/libsvcmi_$V.so; fi
  timeout 300 python scripts/microbench.py gemm biggemm > $OUT/micro_$V.log 2>&1
  echo "--- $V"; grep -E "whisper_(qkv|mlp1).*tile=(0|1) split=1|whisper_(o|mlp2).*tile=1 split=(2|4)|square4096 +f32" $OUT/micro_$V.log | head -12
done
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'), d.get('parity_max_abs_vs_oracle'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
for V in default midbar default midbar; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$EXP/libsvcmi_$V.so; fi
  timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/bench_$V.json 2> $OUT/bench_$V.err; show $OUT/bench_$V.json $V
done
unset SVCMI_LIB
echo "== done"
