#!/bin/bash
# Round 5, GPU session 13: lanes / in-lane batches of configs[3] (fp32, batches of 16) and configs[4] (30 s clips, mixed)
TAG=${1:-r05m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d['config'].get('precision_error') or {}).get('live_max_abs_vs_fp32_engine'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-single-stream $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
for I in 1 2 3 4; do run c3_i$I "--config 3 --inflight $I" X=1; done
run c3_b32_i2 "--config 3 --batch 32 --inflight 2" X=1
run c3_b8_i4 "--config 3 --batch 8 --inflight 4" X=1
run c4_b1_i4 "--config 4" X=1
run c4_b2_i2 "--config 4 --batch 2 --inflight 2 --steps 10" X=1
run c4_b2_i4 "--config 4 --batch 2 --inflight 4 --steps 10" X=1
run c4_b4_i2 "--config 4 --batch 4 --inflight 2 --steps 6" X=1
echo "== done"
