#!/bin/bash
# round-2 GPU session 6: float4 GEMM epilogue
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2j; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_gemm or grouped or splitk or inlaunch" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python scripts/microbench.py gemm 2>&1 | grep -E "tile=(1|6) " | grep -E "split=(1|2|4):" > $OUT/microbench_gemm.log; cat $OUT/microbench_gemm.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json; d = json.load(open("gpurun_out/r2j/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["kernel_time_ms"])
PY
timeout 600 python bench.py --no-cpu-baseline --precision bf16x3 > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err; python - <<'PY'
import json; d = json.load(open("gpurun_out/r2j/bench_bf16x3.json")); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["kernel_time_ms"])
PY
