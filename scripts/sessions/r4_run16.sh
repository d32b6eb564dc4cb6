#!/bin/bash
# round-4 GPU session 16: narrow-stage lp half-step always with split weights: the plain f16 mode against the 1e-3 bar
TAG=${1:-r04s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_precision.py -q -x -s -p no:cacheprovider > $OUT/pytest_prec.log 2>&1; echo "pytest prec rc=$?"; grep -E "configs\[|stress|passed|failed|Error|err " $OUT/pytest_prec.log | tail -40
cp gpurun_out/precision_report.json $OUT/precision_report.json 2>/dev/null
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "err", (d["config"].get("precision_error") or {}).get("live_max_abs_vs_fp32_engine"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --precision f16 --no-cpu-baseline > $OUT/bench_f16.json 2> $OUT/bench_f16.err; show $OUT/bench_f16.json
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2> $OUT/bench_c2_f16.err; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 4 --precision f16 > $OUT/bench_c4_f16.json 2> $OUT/bench_c4_f16.err; show $OUT/bench_c4_f16.json
SVCMI_TUNE="amp_lp=0" timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16_nolp.json 2> $OUT/bench_c2_f16_nolp.err; show $OUT/bench_c2_f16_nolp.json
