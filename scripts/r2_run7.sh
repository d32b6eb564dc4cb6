#!/bin/bash
# round-2 GPU session 7: full validation + the numbers of record after the float4 epilogue
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2k; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "roofline", r.get("achieved"), r.get("frac"), r.get("frac_rocprof"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity_max_abs_vs_oracle"))
PY
}
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; show $OUT/bench.json
for P in bf16x3 bf16 f16; do timeout 600 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; show $OUT/bench_$P.json; done
for C in 2 3 4; do timeout 900 python bench.py --config $C > $OUT/bench_c$C.json 2> $OUT/bench_c$C.err; show $OUT/bench_c$C.json; done
timeout 600 python bench.py --config 2 --precision bf16x3 --no-roofline > $OUT/bench_c2_bf16x3.json 2>/dev/null; show $OUT/bench_c2_bf16x3.json
timeout 600 python bench.py --config 2 --precision f32 --no-roofline > $OUT/bench_c2_f32.json 2>/dev/null; show $OUT/bench_c2_f32.json
timeout 600 python bench.py --config 3 --precision bf16x3 --no-roofline > $OUT/bench_c3_bf16x3.json 2>/dev/null; show $OUT/bench_c3_bf16x3.json
timeout 600 python bench.py --config 4 --precision f32 --no-roofline > $OUT/bench_c4_f32.json 2>/dev/null; show $OUT/bench_c4_f32.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof $OUT/kernel_stats.csv 12 > /dev/null 2>&1; head -24 $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
