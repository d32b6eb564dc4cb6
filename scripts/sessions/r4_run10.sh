#!/bin/bash
# round-4 GPU session 10: run length per thread of the SnakeAlias stream kernel (8 / 12 / 16 outputs) on configs[2] and the judged line
TAG=${1:-r04k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), {k: v for k, v in d.get("kernel_time_ms", {}).items() if "snake_alias" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for RT in 8 12 16; do
  SVCMI_TUNE="snake_rt=$RT" timeout 600 python bench.py --config 2 --steps 10 > $OUT/bench_c2_rt$RT.json 2> $OUT/bench_c2_rt$RT.err; show $OUT/bench_c2_rt$RT.json
  SVCMI_TUNE="snake_rt=$RT" timeout 600 python bench.py --no-cpu-baseline --steps 12 > $OUT/bench_rt$RT.json 2> $OUT/bench_rt$RT.err; show $OUT/bench_rt$RT.json
done
