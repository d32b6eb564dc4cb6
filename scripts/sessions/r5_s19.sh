#!/bin/bash
# Round 5, GPU session 19: the pinned K loop on every tile BUT the 64x64 one (the default build) against the all-tiles form session r05zzzz
# validated (svcmi/exp/libsvcmi_spreadall.so) and, for the single launches, against the build before the pinned loop (libsvcmi_base.so);
# then -- only if the default is not slower than the all-tiles form -- the validation set of record for THIS csrc stamp in the same session.
TAG=${1:-r05zzzzz}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python scripts/spread_check.py $EXP/libsvcmi_base.so > $OUT/spread_check.log 2>&1; echo "spread_check rc=$?"; grep -E "bit comparison|DIFF|^gemm|^group" $OUT/spread_check.log | head -40
lap "spread_check"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run new1 A=1
run all1 SVCMI_LIB=$EXP/libsvcmi_spreadall.so
run new2 A=1
run all2 SVCMI_LIB=$EXP/libsvcmi_spreadall.so
lap "A/B"
DECISION=$(python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
def v(n):
    d = json.loads(open(f"{o}/ab_{n}.json").read().strip().splitlines()[-1]); return d["value"], d["config"]["single_stream"]["value"]
try:
    n = [v("new1"), v("new2")]; b = [v("all1"), v("all2")]
    fl = sum(x[0] for x in n) / sum(x[0] for x in b); ss = sum(x[1] for x in n) / sum(x[1] for x in b)
    print(f"{'go' if fl >= 0.999 and ss >= 0.999 else 'stop'} in-flight {fl:.4f} single-stream {ss:.4f}")
except Exception as e:
    print("stop unreadable", e)
PY
)
echo "decision: $DECISION"
case "$DECISION" in go*) ;; *) echo "== the default build is slower than the all-tiles form (or unreadable): nothing else run"; exit 0;; esac
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
python scripts/traffic_summary.py $OUT/traffic $OUT/traffic.json 3 2>&1 | tail -3
find $OUT -name "*counter_collection.csv" -delete
lap "traffic"
cp $OUT/kernel_stats.csv profiles/${TAG}_kernel_stats.csv; cp $OUT/traffic.json profiles/${TAG}_traffic.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['parity_max_abs_vs_oracle'])"
lap "bench"
ls $OUT
echo "== done"
