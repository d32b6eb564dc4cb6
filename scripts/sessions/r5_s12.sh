#!/bin/bash
# Round 5, GPU session 12: the mixed-precision configs[1] line with clips batched inside a lane (B clips per lane x lanes)
TAG=${1:-r05l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-single-stream --precision mixed"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d['config'].get('precision_error') or {}).get('live_max_abs_vs_fp32_engine'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 $B $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run b1_i4 "--steps 40" X=1
run b2_i4 "--batch 2 --inflight 4 --steps 20" X=1
run b4_i2 "--batch 4 --inflight 2 --steps 12" X=1
run b4_i3 "--batch 4 --inflight 3 --steps 12" X=1
run b4_i4 "--batch 4 --inflight 4 --steps 12" X=1
run b8_i2 "--batch 8 --inflight 2 --steps 8" X=1
run b8_i4 "--batch 8 --inflight 4 --steps 8" X=1
run b16_i2 "--batch 16 --inflight 2 --steps 6" X=1
python bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-single-stream --precision f16 --batch 4 --inflight 4 --steps 12 > $OUT/bench_f16_b4_i4.json 2>/dev/null; show $OUT/bench_f16_b4_i4.json f16_b4_i4
python bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-single-stream --precision bf16x3 --batch 4 --inflight 2 --steps 12 > $OUT/bench_x3_b4_i2.json 2>/dev/null; show $OUT/bench_x3_b4_i2.json bf16x3_b4_i2
echo "== done"
