#!/bin/bash
# Round 5, GPU session 6: configs[2] (B = 16, mixed) under the occupancy-relevant knobs of the narrow half-steps and the SnakeAlias stream kernel
TAG=${1:-r05f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --config 2 --steps 12 --warmup 3 --no-roofline --no-single-stream"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d['config'].get('precision_error') or {}).get('live_max_abs_vs_fp32_engine'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 300 $B $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run c2_default "" X=1
run c2_noU "" SVCMI_TUNE=amp_u=-1
run c2_U "" SVCMI_TUNE=amp_u=1
run c2_rt12 "" SVCMI_TUNE=snake_rt=12
run c2_rt16 "" SVCMI_TUNE=snake_rt=16
run c2_gnst3 "" SVCMI_TUNE=group_nst=3
run c2_i2 "--inflight 2" X=1
run c2_i4 "--inflight 4" X=1
run c2_default_again "" X=1
( time timeout 300 python bench.py --steps 20 --warmup 3 > $OUT/bench_default_timed.json 2> $OUT/bench_default_timed.err ) 2>&1 | grep real
python -c "import json;d=json.loads(open('$OUT/bench_default_timed.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_max_abs_vs_oracle'])"
echo "== done"
