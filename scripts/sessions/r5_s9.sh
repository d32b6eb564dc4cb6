#!/bin/bash
# Round 5, GPU session 9: the U-tile half-steps with their x window staged in LDS (knob amp_x: bit 0 = 10 channels, bit 1 = 20 channels)
TAG=${1:-r05i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "x_window or snake_conv_group" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
for X in 0 3; do
  SVCMI_TUNE=amp_x=$X timeout 300 python scripts/microbench.py amplp > $OUT/micro_amplp_x$X.log 2>&1; echo "--- amp_x=$X"; grep -E "amplp .* d=1 amp_u= 1" $OUT/micro_amplp_x$X.log | sed 's/  max diff.*//'
done
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 python bench.py --no-roofline --no-cpu-baseline $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
for X in 0 1 2 3; do run c2_x$X "--config 2 --steps 12 --warmup 3" SVCMI_TUNE=amp_x=$X; done
for X in 0 1 3; do run f32_x$X "--steps 40" SVCMI_TUNE=amp_x=$X; done
for X in 0 1 3; do run c4_x$X "--config 4" SVCMI_TUNE=amp_x=$X; done
echo "== done"
