#!/bin/bash
# Round 5, last GPU call: the round's HEAD as the driver will run it -- smoke, the default bench command, the in-flight / ring bit-identity tests
TAG=${1:-r05zz_head}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( time timeout 200 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['parity_max_abs_vs_oracle'])"
timeout 100 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "in_flight or two_deep_ring or clip_lanes" > $OUT/pytest_inflight.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_inflight.log
echo "== done"
