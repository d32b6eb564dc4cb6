#!/bin/bash
# split-bf16 on 16-bit activation rows (SVCMI_PREC_BF16X3_A16): kernel/engine tests of the new path, GEMM microbench, bf16x3 bench lines, drop-in times
TAG=${1:-r03p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "reduced_precision or outputs16 or grouped or crepe or bf16x3 or extractors or precision or cpp_host" > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_subset.log
timeout 600 python scripts/microbench.py x3a > $OUT/microbench_x3a.log 2>&1; cat $OUT/microbench_x3a.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), "parity", d.get("parity_max_abs_vs_oracle"))
PY
}
timeout 600 python bench.py --precision bf16x3 --no-cpu-baseline > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err; show $OUT/bench_bf16x3.json
timeout 600 python bench.py --config 3 --precision bf16x3 --no-roofline > $OUT/bench_c3_bf16x3.json 2>/dev/null; show $OUT/bench_c3_bf16x3.json
timeout 600 python bench.py --config 2 --precision bf16x3 --no-roofline > $OUT/bench_c2_bf16x3.json 2>/dev/null; show $OUT/bench_c2_bf16x3.json
timeout 600 python scripts/dropin_times.py 10 bf16x3 > $OUT/dropin_times.log 2>&1; tail -9 $OUT/dropin_times.log
