#!/bin/bash
# Round 6, session 8: the eight-wave tile inside the pipeline (SVCMI_WHISPER_TUNE), 4 clips in flight and one clip at a time.
TAG=${1:-r06p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; env "$@" timeout 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/ab_$name.json 2> $OUT/ab_$name.err; show $OUT/ab_$name.json "$name"; }
run base A=1
run mlp10 SVCMI_WHISPER_TUNE="tile_mlp1=10,tile_mlp2=10"
run mlp10_o10s4 SVCMI_WHISPER_TUNE="tile_mlp1=10,tile_mlp2=10,tile_o=10,split_o=4"
run mlp10_ring2all SVCMI_WHISPER_TUNE="tile_mlp1=10,tile_mlp2=10" SVCMI_TUNE="ring2=12"
run base_ring2all SVCMI_TUNE="ring2=12"
run base2 A=1
run mlp10_2 SVCMI_WHISPER_TUNE="tile_mlp1=10,tile_mlp2=10"
run up10 SVCMI_WHISPER_TUNE="tile_mlp1=10"
run mlp10_ring0 SVCMI_WHISPER_TUNE="tile_mlp1=10,tile_mlp2=10" SVCMI_RING2=0
echo "== done"
