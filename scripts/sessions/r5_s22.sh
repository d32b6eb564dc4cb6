#!/bin/bash
# Round 5, GPU session 22: the 2-deep-ring masks once more, on the pinned K loop (its kernels hold fewer registers: 92 instead of 100 for the
# 64x80 tile) -- SVCMI_RING2 bits: 1 QKV, 2 out-projection, 4 MLP-up, 8 MLP-down, 16 synthesizer GEMMs (svcmi/serving.py: RING2_IN_FLIGHT = 12)
TAG=${1:-r05u}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('gemm_ring2_mask'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run r12 SVCMI_RING2=12
run r15 SVCMI_RING2=15
run r13 SVCMI_RING2=13
run r31 SVCMI_RING2=31
run r0 SVCMI_RING2=0
run r12_again SVCMI_RING2=12
run r15_again SVCMI_RING2=15
echo "== done"
