#!/bin/bash
# round-4 GPU session 8: attention16 with the in-register transposing V staging (ds_write_b128 units): tests, microbench, PMC of an f16 step
TAG=${1:-r04h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "attention" > $OUT/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_attn.log
timeout 600 python scripts/microbench.py attn16 > $OUT/attn16.log 2>&1; echo "attn16 rc=$?"; grep "attn16" $OUT/attn16.log | grep -E "fp32|shape= 0|shape=42|shape=81|shape=21|shape=24" | head -40
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
timeout 600 python bench.py --precision f16 --no-cpu-baseline > $OUT/bench_f16.json 2> $OUT/bench_f16.err; show $OUT/bench_f16.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; show $OUT/bench_c4.json
BENCH_ARGS="--precision f16" bash scripts/pmc_bench.sh $TAG/pmc_f16 > /dev/null 2>&1
python scripts/pmc_summary.py $OUT/pmc_f16 $OUT/pmc_f16.json 2>&1 | grep -E "attention16|conv_gemm_kernel<1,1,0,false,5>" | head -6
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
