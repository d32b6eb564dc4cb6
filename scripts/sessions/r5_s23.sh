#!/bin/bash
# Round 5, GPU session 23 (the last GPU minutes of the round, budgeted to the second): the mid-barrier K loop in the 16-bit-ACTIVATION GEMM
# kernels (SVCMI_GEMM_MIDBAR16 = the default build; the fp32 kernels' object code is byte-identical to the validated build) against the
# previous library (svcmi/exp/libsvcmi_pinned.so): bits + single launches, the f16 line alternating, then -- only on a gain -- the 16-bit GPU
# tests and a fresh stamped kernel-stats / traffic pair + the judged line for THIS csrc stamp.
TAG=${1:-r05zzzza}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
OLD=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_pinned.so
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
timeout 100 python scripts/lp_check.py $OLD > $OUT/lp_check.log 2>&1; echo "lp_check rc=$?"; grep -E "bit comparison|DIFF|^gemm" $OUT/lp_check.log | head -30
lap "lp_check"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d['config'].get('precision_error') or {}).get('live_max_abs_vs_fp32_engine'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 120 python bench.py --precision f16 --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run new1 A=1
run old1 SVCMI_LIB=$OLD
run new2 A=1
run old2 SVCMI_LIB=$OLD
lap "A/B f16 line"
DECISION=$(python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
def v(n):
    return json.loads(open(f"{o}/ab_{n}.json").read().strip().splitlines()[-1])["value"]
try:
    r = (v("new1") + v("new2")) / (v("old1") + v("old2"))
    bits = "0 differ" in open(f"{o}/lp_check.log").read()
    print(f"{'go' if r >= 1.01 and bits else 'stop'} f16 line {r:.4f} bits_identical {bits}")
except Exception as e:
    print("stop unreadable", e)
PY
)
echo "decision: $DECISION"
case "$DECISION" in go*) ;; *) echo "== no gain (or bits differ): nothing else run"; exit 0;; esac
timeout 110 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py -m gpu -q -x -p no:cacheprovider -k "reduced_precision or outputs16 or configs1_end_to_end or f16_whisper_mixed or window_modes" > $OUT/pytest_16bit.log 2>&1; RC=$?; echo "pytest 16-bit subset rc=$RC"; tail -3 $OUT/pytest_16bit.log
lap "pytest subset"
[ $RC -ne 0 ] && { echo "== 16-bit tests failed or timed out: no refresh"; exit 0; }
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof1_bench.json 2> $OUT/prof1.err; echo "rocprof single rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof1 $OUT/kernel_stats.csv 11 > /dev/null 2>&1; head -5 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof1 -name "*kernel_trace.csv" -delete
bash scripts/pmc_traffic.sh $TAG/traffic
python scripts/traffic_summary.py $OUT/traffic $OUT/traffic.json 3 2>&1 | tail -2
find $OUT -name "*counter_collection.csv" -delete
lap "stamped profiles"
cp $OUT/kernel_stats.csv profiles/${TAG}_kernel_stats.csv; cp $OUT/traffic.json profiles/${TAG}_traffic.json
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['parity_max_abs_vs_oracle'])"
lap "bench"
echo "== done"
