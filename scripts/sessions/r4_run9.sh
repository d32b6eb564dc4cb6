#!/bin/bash
# round-4 GPU session 9: clips / batches in flight for configs[2] and configs[4] under the mixed policy
TAG=${1:-r04j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for I in 2 4 5; do timeout 600 python bench.py --config 2 --inflight $I --no-roofline --no-single-stream > $OUT/bench_c2_i$I.json 2> $OUT/bench_c2_i$I.err; show $OUT/bench_c2_i$I.json; done
for I in 3 6 8; do timeout 600 python bench.py --config 4 --inflight $I --no-roofline --no-single-stream > $OUT/bench_c4_i$I.json 2> $OUT/bench_c4_i$I.err; show $OUT/bench_c4_i$I.json; done
