#!/bin/bash
# Round 5, GPU session 3: the one-launch fp32 half-step of the 40- / 80-channel stages (snake_gemm_group_kernel): hardware tests, the
# kernel against the two launches it replaces, and the judged line with it off / on / small tiles / 40 channels only.
# scripts/gpu.sh --timeout 900 -- 'bash scripts/sessions/r5_s3.sh r05c'
TAG=${1:-r05c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 200 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "snake_gemm or two_deep or snake_conv_group or grouped" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 300 python scripts/microbench.py ampgemm > $OUT/micro_ampgemm.log 2>&1; echo "microbench rc=$?"; grep ampgemm $OUT/micro_ampgemm.log
run gemm0 SVCMI_TUNE=amp_gemm=0
run gemm1 SVCMI_TUNE=amp_gemm=1
run gemm2 SVCMI_TUNE=amp_gemm=2
run gemm3 SVCMI_TUNE=amp_gemm=3
run gemm0_again SVCMI_TUNE=amp_gemm=0
run gemm1_again SVCMI_TUNE=amp_gemm=1
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -p no:cacheprovider -k "full_10s or generator or streaming or golden or clip_lanes or in_flight" > $OUT/pytest_engine.log 2>&1; echo "engine pytest rc=$?"; tail -2 $OUT/pytest_engine.log
echo "== done"
