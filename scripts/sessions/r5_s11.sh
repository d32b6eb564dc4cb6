#!/bin/bash
# Round 5, GPU session 11: the LDS-staged fp32 attention (K / V of a head staged once per block, shared by QT query tiles) for the Whisper
# windows WITH CLIPS IN FLIGHT (alone it loses for one T = 500 window: 80-160 blocks cannot fill the chip; in flight other lanes can)
TAG=${1:-r05k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 200 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run base X=1
for C in 21 22 24 41 42 44 81 82; do run lds$C SVCMI_TUNE=attn_lds=$C; done
run q32 SVCMI_TUNE=attn_q32=1
run ns2 SVCMI_TUNE=attn_ns=2
run ns8 SVCMI_TUNE=attn_ns=8
run base_again X=1
echo "== done"
