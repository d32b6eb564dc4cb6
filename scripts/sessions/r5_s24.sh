#!/bin/bash
# Round 5, the last GPU seconds: the pinned K loop with accumulator-file MFMAs (the default build again; probe r05w: 2-3 % faster per launch
# than the architectural-register build on every M = 500 shape, bit-identical) against that build (svcmi/exp/libsvcmi_vgpr.so) on the judged
# line, then fresh stamped kernel stats / traffic and the judged line for THIS stamp.
TAG=${1:-r05zzzzb}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
OLD=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_vgpr.so
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'])" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 100 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run new1 A=1
run old1 SVCMI_LIB=$OLD
run new2 A=1
lap "A/B"
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof1_bench.json 2> $OUT/prof1.err; echo "rocprof single rc=$?"
cd $ROOT
python scripts/prof_summary.py $OUT/prof1 $OUT/kernel_stats.csv 11 > /dev/null 2>&1; head -4 $OUT/kernel_stats.csv | cut -c1-160
find $OUT/prof1 -name "*kernel_trace.csv" -delete
bash scripts/pmc_traffic.sh $TAG/traffic
python scripts/traffic_summary.py $OUT/traffic $OUT/traffic.json 3 2>&1 | tail -1
find $OUT -name "*counter_collection.csv" -delete
lap "stamped profiles"
cp $OUT/kernel_stats.csv profiles/${TAG}_kernel_stats.csv; cp $OUT/traffic.json profiles/${TAG}_traffic.json
timeout 100 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default', d['value'], d['ms_per_step'], d['config']['single_stream'], r['frac'], r.get('frac_rocprof'), r.get('traffic'), r.get('stale'), d['cpu_baseline']['value'], d['parity_max_abs_vs_oracle'])"
lap "bench"
echo "== done"
