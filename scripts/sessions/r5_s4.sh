#!/bin/bash
# Round 5, GPU session 4: the judged line with clips batched inside a lane (B x lanes) on top of the 2-deep ring, and the ring mask refined.
# scripts/gpu.sh --timeout 900 -- 'bash scripts/sessions/r5_s4.sh r05d'
TAG=${1:-r05d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'])" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 $B $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run b1_i4_ring12 "--steps 40" SVCMI_RING2=12
run b1_i4_ring4 "--steps 40" SVCMI_RING2=4
run b1_i4_ring8 "--steps 40" SVCMI_RING2=8
run b1_i4_ring14 "--steps 40" SVCMI_RING2=14
run b1_i4_ring13 "--steps 40" SVCMI_RING2=13
run b2_i2_ring12 "--steps 20 --batch 2 --inflight 2" SVCMI_RING2=12
run b2_i3_ring12 "--steps 20 --batch 2 --inflight 3" SVCMI_RING2=12
run b2_i3_ring0 "--steps 20 --batch 2 --inflight 3" SVCMI_RING2=0
run b4_i2_ring12 "--steps 12 --batch 4 --inflight 2" SVCMI_RING2=12
run b4_i2_ring0 "--steps 12 --batch 4 --inflight 2" SVCMI_RING2=0
run b1_i4_ring12_again "--steps 40" SVCMI_RING2=12
echo "== done"
