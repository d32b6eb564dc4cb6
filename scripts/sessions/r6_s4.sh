#!/bin/bash
# Round 6, session 4: the refill spread over a whole K-step (SPREAD2) and the LDS-DMA without the M0 save / restore (M0_RAW), each and both,
# against the shipped library: same bits?  per-launch times; the judged line A/B (4 clips in flight) and one clip at a time.
TAG=${1:-r06f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp; BASE=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so
for v in spread2 m0raw spread2m0; do
  echo "== $v vs shipped"
  SVCMI_LIB=$EXP/libsvcmi_$v.so timeout 400 python scripts/spread_check.py $BASE 2>&1 | grep -v amdgpu.ids | tee $OUT/check_$v.log | grep -v "^group" | tail -n 12
done
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run base1 A=1
run spread2 SVCMI_LIB=$EXP/libsvcmi_spread2.so
run m0raw SVCMI_LIB=$EXP/libsvcmi_m0raw.so
run spread2m0 SVCMI_LIB=$EXP/libsvcmi_spread2m0.so
run base2 A=1
run spread2m0_b SVCMI_LIB=$EXP/libsvcmi_spread2m0.so
echo "== done"
