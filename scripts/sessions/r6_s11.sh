#!/bin/bash
# Round 6, session 11: clips-in-flight sweep at the round-6 kernels (the lanes x queues table was last swept in round 2), configs[1] and configs[3].
TAG=${1:-r06v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("gemm_ring2_mask"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for Q in 8 16; do for N in 2 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --inflight $N --steps 48 --warmup 8 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/c1_q${Q}_n$N.json 2> $OUT/c1_q${Q}_n$N.err; show $OUT/c1_q${Q}_n$N.json "c1 queues=$Q lanes=$N"
done; done
for N in 1 2 3 4; do
  timeout 600 python bench.py --config 3 --inflight $N --no-cpu-baseline --no-roofline > $OUT/c3_n$N.json 2> $OUT/c3_n$N.err; show $OUT/c3_n$N.json "c3 lanes=$N"
done
echo "== done"
