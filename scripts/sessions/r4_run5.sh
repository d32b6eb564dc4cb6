#!/bin/bash
# round-4 GPU session 5: mixed policy with the prior encoder's attention on the 16-bit matrix cores; configs[2] / [4] lines
TAG=${1:-r04e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_precision.py -q -x -s -p no:cacheprovider -k "mixed" > $OUT/pytest_prec.log 2>&1; echo "pytest prec rc=$?"; grep -E "configs\[|stress|passed|failed|Error" $OUT/pytest_prec.log | tail -40
cp gpurun_out/precision_report.json $OUT/precision_report.json 2>/dev/null
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), "err", d["config"].get("precision_error"))
    print("   kernel_time_ms", d.get("kernel_time_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; show $OUT/bench_c2.json
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2> $OUT/bench_c2_f16.err; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 2 --precision "mixed:amp1=bf16x3" > $OUT/bench_c2_amp1.json 2> $OUT/bench_c2_amp1.err; show $OUT/bench_c2_amp1.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; show $OUT/bench_c4.json
timeout 600 python bench.py --config 4 --precision f16 > $OUT/bench_c4_f16.json 2> $OUT/bench_c4_f16.err; show $OUT/bench_c4_f16.json
