#!/bin/bash
# round-4 GPU session 11: fp16 activations x split fp16 weights (f16w2): kernel cases, the policy sweep, configs[2] / [4] lines
TAG=${1:-r04l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "reduced_precision or grouped" > $OUT/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -2 $OUT/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_precision.py -q -x -s -p no:cacheprovider -k "mixed" > $OUT/pytest_prec.log 2>&1; echo "pytest prec rc=$?"; grep -E "configs\[|stress|passed|failed|Error" $OUT/pytest_prec.log | tail -30
cp gpurun_out/precision_report.json $OUT/precision_report.json 2>/dev/null
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "err", (d["config"].get("precision_error") or {}).get("live_max_abs_vs_fp32_engine"))
    print("   kernel_time_ms", {k: v for k, v in d.get("kernel_time_ms", {}).items() if "lp" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for P in "mixed" "mixed:amp0=f16w2" "mixed:amp0=f16w2,amp1=f16w2,amp2=f16w2" "f16w2"; do
  F=$(echo $P | tr ':=,' '___'); timeout 600 python bench.py --config 2 --precision "$P" > $OUT/bench_c2_$F.json 2> $OUT/bench_c2_$F.err; show $OUT/bench_c2_$F.json
done
timeout 600 python bench.py --config 4 --precision "mixed:amp0=f16w2" > $OUT/bench_c4_w2.json 2> $OUT/bench_c4_w2.err; show $OUT/bench_c4_w2.json
