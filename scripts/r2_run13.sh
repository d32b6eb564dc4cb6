#!/bin/bash
# round-2 GPU session 13: full validation + the numbers of record with clips in flight + rocprof summaries (single stream and in flight)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2s; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), r.get("frac_rocprof"), r.get("in_flight", {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity_max_abs_vs_oracle"))
PY
}
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; show $OUT/bench.json
for P in bf16x3 bf16 f16; do timeout 600 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; show $OUT/bench_$P.json; done
for C in 2 3 4; do timeout 900 python bench.py --config $C > $OUT/bench_c$C.json 2> $OUT/bench_c$C.err; show $OUT/bench_c$C.json; done
timeout 600 python bench.py --config 3 --precision bf16x3 --no-roofline > $OUT/bench_c3_bf16x3.json 2>/dev/null; show $OUT/bench_c3_bf16x3.json
timeout 600 python bench.py --config 3 --inflight 1 --no-roofline > $OUT/bench_c3_inflight1.json 2>/dev/null; show $OUT/bench_c3_inflight1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof1_bench.json 2> $OUT/prof1.err; echo "rocprof single rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o trace -- python $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/prof4_bench.json 2> $OUT/prof4.err; echo "rocprof inflight rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof1 $OUT/kernel_stats.csv 11 > /dev/null 2>&1; head -16 $OUT/kernel_stats.csv
python scripts/inflight_summary.py $OUT/prof4 $OUT/inflight4_trace_summary.csv 22 2>&1 | head -20
find $OUT/prof1 $OUT/prof4 -name "*kernel_trace.csv" -delete
