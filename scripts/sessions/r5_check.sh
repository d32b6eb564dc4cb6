#!/bin/bash
# Round 5: sanity session after host-side (Python) changes -- the GPU suite, smoke and the judged line at HEAD
TAG=${1:-r05y}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_max_abs_vs_oracle'])"
