#!/bin/bash
# Round 5, GPU session 18: the pinned K loop (SVCMI_GEMM_SPREAD, conv_gemm.hip with its accumulators in architectural registers = the
# default build) against the library built from the previous commit's sources (svcmi/exp/libsvcmi_base.so):
#   1. same bits on hardware + single-launch timings (scripts/spread_check.py),  2. the judged line, alternating, two runs each;
#   then -- ONLY if the new build is not slower -- the validation set of record in the same session: GPU suite, smoke, the judged line with
#   its CPU baseline, rocprofv3 kernel stats of the single-stream run, PMC HBM traffic (the files bench.py quotes must carry this csrc's stamp).
TAG=${1:-r05s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
BASE=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_base.so
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python scripts/spread_check.py $BASE > $OUT/spread_check.log 2>&1; echo "spread_check rc=$?"; grep -E "bit comparison|DIFF|^gemm|^group" $OUT/spread_check.log | head -40
lap "spread_check"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run new1 A=1
run base1 SVCMI_LIB=$BASE
run new2 A=1
run base2 SVCMI_LIB=$BASE
lap "A/B"
DECISION=$(python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
def v(n):
    d = json.loads(open(f"{o}/ab_{n}.json").read().strip().splitlines()[-1]); return d["value"], d["config"]["single_stream"]["value"]
try:
    n = [v("new1"), v("new2")]; b = [v("base1"), v("base2")]
    fl = sum(x[0] for x in n) / sum(x[0] for x in b); ss = sum(x[1] for x in n) / sum(x[1] for x in b)
    print(f"{'go' if fl >= 0.999 and ss >= 0.999 else 'stop'} in-flight {fl:.4f} single-stream {ss:.4f}")
except Exception as e:
    print("stop unreadable", e)
PY
)
echo "decision: $DECISION"
case "$DECISION" in go*) ;; *) echo "== new build is slower or unreadable: nothing else run"; exit 0;; esac
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
lap "pytest"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof1_bench.json 2> $OUT/prof1.err; echo "rocprof single rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof1 $OUT/kernel_stats.csv 11 > /dev/null 2>&1; head -12 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof1 -name "*kernel_trace.csv" -delete
lap "kernel stats"
bash scripts/pmc_traffic.sh $TAG/traffic
python scripts/traffic_summary.py $OUT/traffic $OUT/traffic.json 3 2>&1 | tail -4
find $OUT -name "*counter_collection.csv" -delete
lap "traffic"
# the judged line last: with this session's own stamped profile files in place (the driver's run will read the committed copies)
cp $OUT/kernel_stats.csv profiles/${TAG}_kernel_stats.csv; cp $OUT/traffic.json profiles/${TAG}_traffic.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['parity_max_abs_vs_oracle'])"
lap "bench"
# expendable last step: what the chip does with 4 clips in flight (kernel trace of the in-flight run, scripts/inflight_trace.py)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof4 -o trace -- python $ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/prof4_bench.json 2> $OUT/prof4.err; echo "rocprof in-flight rc=$?"
cd $ROOT
python scripts/inflight_trace.py $OUT/prof4 > $OUT/inflight_trace.txt 2>&1; head -12 $OUT/inflight_trace.txt
find $OUT/prof4 -name "*.csv" -delete
lap "in-flight trace"
ls $OUT
echo "== done"
