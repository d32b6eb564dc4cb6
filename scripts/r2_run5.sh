#!/bin/bash
# round-2 GPU session 5: state after the LP heuristics / threshold / merged waits
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2h; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py -m gpu -q -x -k "conv_gemm or grouped or configs1 or whisper_15s" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "roofline", r.get("achieved"), r.get("frac"), {k: v for k, v in list(d.get("kernel_time_ms", {}).items())[:6]})
PY
}
for P in f32 bf16x3 bf16 f16; do timeout 600 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; show $OUT/bench_$P.json; done
timeout 600 python bench.py --config 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json
timeout 600 python bench.py --config 3 --precision bf16x3 --no-roofline > $OUT/bench_c3_bf16x3.json 2> $OUT/bench_c3_bf16x3.err; show $OUT/bench_c3_bf16x3.json
timeout 600 python bench.py --config 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; show $OUT/bench_c4.json
timeout 600 python bench.py --config 2 --precision bf16x3 > $OUT/bench_c2_bf16x3.json 2> $OUT/bench_c2_bf16x3.err; show $OUT/bench_c2_bf16x3.json
