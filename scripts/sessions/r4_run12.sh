#!/bin/bash
# round-4 GPU session 12: the narrow stages' half-step on the fp16 matrix cores (svcmi_snake_conv_group_lp): kernel cases, microbench
# against the fp32 vector kernel, configs[2] / [4] lines with and without it
TAG=${1:-r04o}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake_conv_group" > $OUT/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $OUT/pytest_k.log
timeout 600 python scripts/microbench.py amplp > $OUT/amplp.log 2>&1; echo "amplp rc=$?"; grep amplp $OUT/amplp.log
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
timeout 600 python bench.py --config 2 --precision mixed > $OUT/bench_c2_mixed.json 2> $OUT/bench_c2_mixed.err; show $OUT/bench_c2_mixed.json
SVCMI_TUNE="amp_lp=0" timeout 600 python bench.py --config 2 --precision mixed > $OUT/bench_c2_mixed_nolp.json 2> $OUT/bench_c2_mixed_nolp.err; show $OUT/bench_c2_mixed_nolp.json
timeout 600 python bench.py --config 2 --precision "mixed:amp3=f16w2,amp4=f16w2" > $OUT/bench_c2_mixed_w2.json 2> $OUT/bench_c2_mixed_w2.err; show $OUT/bench_c2_mixed_w2.json
timeout 600 python bench.py --config 2 --precision "mixed:amp3=f16,amp4=f16w2" > $OUT/bench_c2_mixed_a4w2.json 2> $OUT/bench_c2_mixed_a4w2.err; show $OUT/bench_c2_mixed_a4w2.json
timeout 600 python bench.py --config 4 --precision mixed > $OUT/bench_c4_mixed.json 2> $OUT/bench_c4_mixed.err; show $OUT/bench_c4_mixed.json
SVCMI_TUNE="amp_lp=0" timeout 600 python bench.py --config 4 --precision mixed > $OUT/bench_c4_mixed_nolp.json 2> $OUT/bench_c4_mixed_nolp.err; show $OUT/bench_c4_mixed_nolp.json
