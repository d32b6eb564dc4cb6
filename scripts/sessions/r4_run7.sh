#!/bin/bash
# round-4 GPU session 7: relative-position attention through the LDS-staged kernel
TAG=${1:-r04g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python scripts/microbench.py attnrel > $OUT/attnrel.log 2>&1; echo "rc=$?"; grep attnrel $OUT/attnrel.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "attention" > $OUT/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_attn.log
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "err", (d["config"].get("precision_error") or {}).get("live_max_abs_vs_fp32_engine"))
    print("   kernel_time_ms", {k: v for k, v in d.get("kernel_time_ms", {}).items() if "attention" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --config 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json
SVCMI_TUNE="attn_lds=-1" timeout 600 python bench.py --config 2 > $OUT/bench_c2_nolds.json 2> $OUT/bench_c2_nolds.err; show $OUT/bench_c2_nolds.json
timeout 900 python bench.py --config 3 --no-roofline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json
