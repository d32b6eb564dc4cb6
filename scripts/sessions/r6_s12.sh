#!/bin/bash
# Round 6, session 12: (i) the reduced-precision configs[1] lines after the Ops lock became re-entrant (r06z: bench.py --precision f16 | bf16x3 | mixed
# deadlocked inside ClipLanes.capture -- the capture's warm-up builds the 16-bit weight images under the same lock), (ii) batched Whisper windows
# as ONE M = B * tw matrix per projection (configs[3]).
TAG=${1:-r06w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider -k "clip_lanes or batch16 or pred_ppg or whisper or config0" > $OUT/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.log
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), "err", d["config"].get("precision_error"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for P in f16 bf16x3 mixed; do timeout 400 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; echo "bench $P rc=$?"; show $OUT/bench_$P.json; done
timeout 900 python bench.py --config 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 rc=$?"; show $OUT/bench_c3.json
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_c1.json 2> $OUT/bench_c1.err; show $OUT/bench_c1.json
echo "== done"
