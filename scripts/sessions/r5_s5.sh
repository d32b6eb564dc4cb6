#!/bin/bash
# Round 5, GPU session 5: lanes sweep of the 16-bit configs[1] lines (latency-bound kernels: do more clips in flight help?)
TAG=${1:-r05e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d['config'].get('precision_error') or {}).get('live_max_abs_vs_fp32_engine'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 $B $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
for P in mixed f16; do
  for L in 4 6 8; do run ${P}_i$L "--precision $P --inflight $L" GPU_MAX_HW_QUEUES=16; done
done
run mixed_i6_q8 "--precision mixed --inflight 6" GPU_MAX_HW_QUEUES=8
run mixed_b2_i3 "--precision mixed --batch 2 --inflight 3 --steps 20" GPU_MAX_HW_QUEUES=16
run mixed_b2_i4 "--precision mixed --batch 2 --inflight 4 --steps 20" GPU_MAX_HW_QUEUES=16
run f32_i4_q16 "" GPU_MAX_HW_QUEUES=16
echo "== done"
