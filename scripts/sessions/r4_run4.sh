#!/bin/bash
# round-4 GPU session 4: clips batched inside a lane (B clips per Whisper / synthesizer pass) x lanes
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
export SVCMI_TUNE="amp_block=0"
for BI in "2 1" "2 2" "2 3" "4 1" "4 2" "3 2" "8 1"; do
  set -- $BI
  timeout 600 python bench.py --no-cpu-baseline --no-roofline --batch $1 --inflight $2 --steps 12 > $OUT/bench_b$1_i$2.json 2> $OUT/bench_b$1_i$2.err; show $OUT/bench_b$1_i$2.json
done
