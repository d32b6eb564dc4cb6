#!/bin/bash
# round-4 GPU session 3: SnakeAlias arithmetic variants (one large-argument test per work item: all chains interleavable / 4 at a time / the
# round-3 per-pair form) x (half-step chain, fused block kernel, SnakeAlias stream kernel), then the judged line with the best
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for V in default c4 pp; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so; fi
  timeout 600 python scripts/microbench.py ampblock quick snake > $OUT/micro_$V.log 2>&1; echo "== $V rc=$?"; grep -E "ampblock|snake C" $OUT/micro_$V.log
done
unset SVCMI_LIB
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"))
    print("   kernel_time_ms", d.get("kernel_time_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for V in default c4 pp; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so; fi
  SVCMI_TUNE="amp_block=0" timeout 600 python bench.py --no-cpu-baseline --steps 12 > $OUT/bench_$V.json 2> $OUT/bench_$V.err; show $OUT/bench_$V.json
done
