#!/bin/bash
# round-4 GPU session 6: SnakeWindow pairs + folded up-sampler gain + saddr interior loads in the stream kernel (default build) vs HEAD
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"))
    print("   kernel_time_ms", {k: v for k, v in d.get("kernel_time_ms", {}).items() if "snake" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for V in default head; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so; fi
  timeout 600 python scripts/microbench.py snake ampgroup > $OUT/micro_$V.log 2>&1; echo "== $V rc=$?"; grep -E "snake C|ampgroup" $OUT/micro_$V.log | head -20
  timeout 600 python bench.py --no-cpu-baseline --steps 12 > $OUT/bench_$V.json 2> $OUT/bench_$V.err; show $OUT/bench_$V.json
  timeout 600 python bench.py --config 2 --steps 8 > $OUT/bench_c2_$V.json 2> $OUT/bench_c2_$V.err; show $OUT/bench_c2_$V.json
done
