#!/bin/bash
# Round 5, GPU session 16: the 2-deep ring re-examined on top of the mid-barrier K loop (a refilled slot now has NST-1 full K-steps of lead)
TAG=${1:-r05q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run r12 "" SVCMI_RING2=12
run r0 "" SVCMI_RING2=0
run r31 "" SVCMI_RING2=31
run r15 "" SVCMI_RING2=15
run r13 "" SVCMI_RING2=13
run one_r0 "--inflight 1 --no-single-stream" SVCMI_RING2=0
run one_r15 "--inflight 1 --no-single-stream" SVCMI_RING2=15
run one_r31 "--inflight 1 --no-single-stream" SVCMI_RING2=31
run r12_again "" SVCMI_RING2=12
echo "== done"
