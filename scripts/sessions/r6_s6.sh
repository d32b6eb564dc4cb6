#!/bin/bash
# Round 6, session 6: the load-free epilogue loop (no s_waitcnt vmcnt(0) per iteration): hardware tests, epilogue cycles, judged line A/B.
TAG=${1:-r06k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or gemm or gelu or layernorm" 2>&1 | tail -n 3 | tee $OUT/pytest_kernels.log
timeout 600 python scripts/spread_check.py $EXP/libsvcmi_m0raw.so 2>&1 | grep -v amdgpu.ids | tee $OUT/check_new_vs_old_epilogue.log | grep -v "^group" | tail -n 10
for v in kt5120 kt3840; do
  SVCMI_KTRACE_CLOCK=1 SVCMI_LIB=$EXP/libsvcmi_$v.so timeout 200 python bench.py --inflight 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/${v}_single.json 2> $OUT/${v}_single.err; grep "ktrace sample" $OUT/${v}_single.err | tail -n 2
done
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run old_m0raw SVCMI_LIB=$EXP/libsvcmi_m0raw.so
run new A=1
run full SVCMI_LIB=$EXP/libsvcmi_full.so
run old_m0raw2 SVCMI_LIB=$EXP/libsvcmi_m0raw.so
run new2 A=1
run full2 SVCMI_LIB=$EXP/libsvcmi_full.so
echo "== microbench wtune (the split-K slab launches), new then old"
timeout 300 python scripts/microbench.py wtune 2>&1 | grep -v amdgpu.ids | grep "T=500" | grep "whisper_o\|whisper_mlp2" | tee $OUT/wtune_new.log
SVCMI_LIB=$EXP/libsvcmi_m0raw.so timeout 300 python scripts/microbench.py wtune 2>&1 | grep -v amdgpu.ids | grep "T=500" | grep "whisper_o\|whisper_mlp2" | tee $OUT/wtune_old.log
echo "== done"
