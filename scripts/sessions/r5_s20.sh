#!/bin/bash
# Round 5, GPU session 20: the OTHER lines of the set of record at the shipped build (the pinned K loop, csrc stamp of r05zzzz): the reduced-
# precision forms of configs[1], configs[2] / [3] / [4], the 2-rank plumbing run on one GPU, and the K / M scaling of the Whisper tiles.
# (The judged line, the GPU suite, smoke and the stamped rocprofv3 / PMC summaries of this build are session r05zzzz, scripts/sessions/r5_s18.sh.)
TAG=${1:-r05zzzz}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), "err", d["config"].get("precision_error"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for C in 2 4; do timeout 600 python bench.py --config $C > $OUT/bench_c$C.json 2> $OUT/bench_c$C.err; show $OUT/bench_c$C.json; done
lap "c2 c4"
for P in mixed f16 bf16x3; do timeout 600 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; show $OUT/bench_$P.json; done
lap "configs[1] reduced precision"
timeout 600 python bench.py --config 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json
lap "c3"
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2>/dev/null; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 2 --precision "mixed:encattn=f16" > $OUT/bench_c2_encattn16.json 2>/dev/null; show $OUT/bench_c2_encattn16.json
timeout 600 python bench.py --config 4 --precision f16 > $OUT/bench_c4_f16.json 2>/dev/null; show $OUT/bench_c4_f16.json
lap "f16 variants"
SVCMI_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-roofline > $OUT/bench_2ranks_1gpu.json 2> $OUT/bench_2ranks_1gpu.err; echo "2 ranks (self-spawned, gloo, one GPU) rc=$?"; show $OUT/bench_2ranks_1gpu.json
lap "2 ranks"
timeout 200 python scripts/microbench.py kscale > $OUT/kscale.log 2>&1; grep -E "^gemm" $OUT/kscale.log
lap "kscale"
echo "== done"
