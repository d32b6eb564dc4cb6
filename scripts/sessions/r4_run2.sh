#!/bin/bash
# round-4 GPU session 2: fused AMP-block kernel after the scalar-load fix + benches
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python scripts/microbench.py ampblock > $OUT/ampblock.log 2>&1; echo "ampblock rc=$?"; grep ampblock $OUT/ampblock.log | tail -40
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
if [ "$2" != "micro" ]; then
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; show $OUT/bench.json
SVCMI_TUNE="amp_block=0" timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_noampblock.json 2> $OUT/bench_noampblock.err; show $OUT/bench_noampblock.json
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; show $OUT/bench_c2.json
SVCMI_TUNE="amp_block=0" timeout 600 python bench.py --config 2 > $OUT/bench_c2_noampblock.json 2> $OUT/bench_c2_noampblock.err; show $OUT/bench_c2_noampblock.json
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2> $OUT/bench_c2_f16.err; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; show $OUT/bench_c4.json
fi
