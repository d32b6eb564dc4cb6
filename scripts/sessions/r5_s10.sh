#!/bin/bash
# Round 5, GPU session 10: Whisper tile / split table re-swept with 4 clips in flight ON TOP of the 2-deep ring (LDS residency is what the
# in-flight step responds to): 64x64 tiles (33 KB with ring2 = 4 blocks per CU) for the MLP GEMMs, fewer K slices for the two N = 1280 projections.
TAG=${1:-r05j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'])" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 200 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run base X=1
run m1_64x64_r12 SVCMI_WHISPER_TUNE=tile_mlp1=1
run m12_64x64_r12 SVCMI_WHISPER_TUNE=tile_mlp1=1,tile_mlp2=1
run m12_64x64_r15 SVCMI_WHISPER_TUNE=tile_mlp1=1,tile_mlp2=1 SVCMI_RING2=15
run all_64x64_r15 SVCMI_WHISPER_TUNE=tile_mlp1=1,tile_mlp2=1,tile_o=1 SVCMI_RING2=15
run qkv_64x80_r13 SVCMI_WHISPER_TUNE=tile_qkv=6 SVCMI_RING2=13
run qkv_64x80_r15 SVCMI_WHISPER_TUNE=tile_qkv=6 SVCMI_RING2=15
run so1 SVCMI_WHISPER_TUNE=split_o=1
run so1_sm2 SVCMI_WHISPER_TUNE=split_o=1,split_mlp=2
run so1_sm1 SVCMI_WHISPER_TUNE=split_o=1,split_mlp=1
run sm2 SVCMI_WHISPER_TUNE=split_mlp=2
run sm8 SVCMI_WHISPER_TUNE=split_mlp=8
run base_again X=1
echo "== done"
