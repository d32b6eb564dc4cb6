#!/bin/bash
# round-4 GPU session 14: lp half-step as shipped (U tile for both widths, amp3 / amp4 = f16w2 by default): precision sweep + lines
TAG=${1:-r04q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake_conv_group" > $OUT/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -2 $OUT/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_precision.py -q -x -s -p no:cacheprovider -k "mixed" > $OUT/pytest_prec.log 2>&1; echo "pytest prec rc=$?"; grep -E "configs\[|stress|passed|failed|Error" $OUT/pytest_prec.log | tail -30
cp gpurun_out/precision_report.json $OUT/precision_report.json 2>/dev/null
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "err", (d["config"].get("precision_error") or {}).get("live_max_abs_vs_fp32_engine"))
    print("   kernel_time_ms", {k: v for k, v in d.get("kernel_time_ms", {}).items() if "snake" in k or "lp" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; show $OUT/bench_c4.json
timeout 600 python bench.py --config 2 --precision "mixed:amp3=f16,amp4=f16" > $OUT/bench_c2_f16.json 2> $OUT/bench_c2_f16.err; show $OUT/bench_c2_f16.json
