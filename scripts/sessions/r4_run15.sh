#!/bin/bash
# round-4 GPU session 15: the fp32 half-step with its convolution on the fp32 matrix cores (snake_convm_group_kernel): kernel cases,
# microbench against the vector-ALU kernel, engine parity, the judged line with and without it
TAG=${1:-r04r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake_conv" > $OUT/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -2 $OUT/pytest_k.log
timeout 600 python scripts/microbench.py amplp > $OUT/amplp.log 2>&1; echo "amplp rc=$?"; grep "amplp" $OUT/amplp.log | grep -v "f16"
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -p no:cacheprovider -k "full_10s or generator or streaming or ungrouped or cpp_host" > $OUT/pytest_e.log 2>&1; echo "pytest engine rc=$?"; tail -2 $OUT/pytest_e.log
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "parity", d.get("parity_max_abs_vs_oracle"), "frac", r.get("frac"))
    print("   kernel_time_ms", {k: v for k, v in d.get("kernel_time_ms", {}).items() if "snake" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; show $OUT/bench.json
SVCMI_TUNE="amp_mfma=0" timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_valu.json 2> $OUT/bench_valu.err; show $OUT/bench_valu.json
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; show $OUT/bench2.json
SVCMI_TUNE="amp_mfma=0" timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_valu2.json 2> $OUT/bench_valu2.err; show $OUT/bench_valu2.json
timeout 600 python bench.py --config 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json
