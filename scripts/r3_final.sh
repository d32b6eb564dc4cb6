#!/bin/bash
# round-3 validation: full GPU test suite, smoke, the judged bench line and the other BASELINE.json configurations, drop-in timings.
TAG=${1:-r03m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d["dtype"], "value", d["value"], "ms", d["ms_per_step"], "single", d["config"].get("single_stream"), "roofline", r.get("achieved"), r.get("frac"), r.get("frac_rocprof"), r.get("in_flight", {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity_max_abs_vs_oracle"), "err", (d["config"].get("precision_error") or {}).get("waveform_max_abs_err_vs_fp32_oracle"))
PY
}
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; show $OUT/bench.json
for P in bf16x3 bf16 f16; do timeout 600 python bench.py --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err; show $OUT/bench_$P.json; done
for C in 2 3 4; do timeout 900 python bench.py --config $C > $OUT/bench_c$C.json 2> $OUT/bench_c$C.err; show $OUT/bench_c$C.json; done
timeout 600 python bench.py --config 2 --precision f16 > $OUT/bench_c2_f16.json 2>/dev/null; show $OUT/bench_c2_f16.json
timeout 600 python bench.py --config 3 --precision bf16x3 --no-roofline > $OUT/bench_c3_bf16x3.json 2>/dev/null; show $OUT/bench_c3_bf16x3.json
timeout 600 python scripts/dropin_times.py 10 bf16x3 > $OUT/dropin_times.log 2>&1; tail -9 $OUT/dropin_times.log
timeout 600 python scripts/dropin_times.py 10 f16 > $OUT/dropin_times_f0_f16.log 2>&1; tail -3 $OUT/dropin_times_f0_f16.log      # --f0-precision f16 (CREPE on the 16-bit-activation kernels)
# two ranks on the one GPU (gloo: RCCL refuses duplicate devices): the N > 1 code path of bench.py (packed broadcast -> views -> C model structs)
SVCMI_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-roofline > $OUT/bench_2ranks_1gpu.json 2> $OUT/bench_2ranks_1gpu.err; echo "2 ranks rc=$?"; show $OUT/bench_2ranks_1gpu.json
# rocprofv3 kernel stats + PMC of the f16 line (16-bit activations, f16 attention)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f16 -o trace -- python $ROOT/bench.py --precision f16 --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_f16_bench.json 2> $OUT/prof_f16.err; echo "rocprof f16 rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof_f16 $OUT/kernel_stats_f16.csv 11 > /dev/null 2>&1; head -16 $OUT/kernel_stats_f16.csv
find $OUT/prof_f16 -name "*kernel_trace.csv" -delete
BENCH_ARGS="--precision f16" bash scripts/pmc_bench.sh $TAG/pmc_f16
python scripts/pmc_summary.py $OUT/pmc_f16 $OUT/pmc_f16.json 2>&1 | tail -6
find $OUT -name "*counter_collection.csv" -delete
