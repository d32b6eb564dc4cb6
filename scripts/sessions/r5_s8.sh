#!/bin/bash
# Round 5, GPU session 8: the residual prefetch in the vector-ALU half-step too (baseline without it: r05g)
TAG=${1:-r05h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "snake_conv or fp16_matrix or streaming or snake_post" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 300 python scripts/microbench.py amplp > $OUT/micro_amplp.log 2>&1; grep -E "amplp .* d=1 amp_u= 1" $OUT/micro_amplp.log | sed 's/  max diff.*//'
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 python bench.py --no-roofline --no-cpu-baseline $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run f32 "--steps 40" X=1
run f32_again "--steps 40" X=1
run c3 "--config 3" X=1
echo "== done"
