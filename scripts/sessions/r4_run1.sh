#!/bin/bash
# round-4 GPU session 1: fused AMP-block kernel (variants vs the half-step chain), mixed-precision policy sweep, judged line A/B
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python scripts/microbench.py ampblock > $OUT/ampblock.log 2>&1; echo "ampblock rc=$?"; cat $OUT/ampblock.log | tail -40
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "amp_block or snake" > $OUT/pytest_amp.log 2>&1; echo "pytest amp rc=$?"; tail -3 $OUT/pytest_amp.log
timeout 900 python -m pytest tests/test_gpu_precision.py -q -x -s -p no:cacheprovider -k "mixed or configs2" > $OUT/pytest_prec.log 2>&1; echo "pytest prec rc=$?"; grep -E "configs\[|stress|passed|failed|Error" $OUT/pytest_prec.log | tail -40
cp gpurun_out/precision_report.json $OUT/precision_report.json 2>/dev/null
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), "err", d["config"].get("precision_error"), "parity", d.get("parity_max_abs_vs_oracle"))
    print("   kernel_time_ms", d.get("kernel_time_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; show $OUT/bench.json
SVCMI_TUNE="amp_block=0" timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_noampblock.json 2> $OUT/bench_noampblock.err; show $OUT/bench_noampblock.json
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; show $OUT/bench_c2.json
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2> $OUT/bench_c2_f16.err; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; show $OUT/bench_c4.json
