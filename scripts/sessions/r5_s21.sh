#!/bin/bash
# Round 5, GPU session 21: the operand-fill ceiling.  Session r05zzzz's K / M scaling (kscale) puts every 64-row fp32 tile at the same
# ~6.3 TB/s of global -> LDS fill (10.6 B per clock and CU: the chip's streaming rate), whatever the occupancy -- so with clips in flight what
# counts is fill bytes per FLOP, not blocks per launch.  Here: the Whisper window GEMMs on the 128-row tiles of the SAME micro-kernel policy
# (P16 128x80 instead of 64x80: -28 % fill per FLOP; 128x64 instead of 64x64: -25 %; same K order = same bits), 4 clips in flight.
TAG=${1:-r05t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config'].get('single_stream'), d.get('parity_max_abs_vs_oracle'))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
run() { local name=$1; shift; local args=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"; }
run base "--no-single-stream" A=1
run m1_7 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_mlp1=7
run m12_7 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_mlp1=7,tile_mlp2=7
run m12o_7 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_mlp1=7,tile_mlp2=7,tile_o=7
run qkv_2 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_qkv=2
run qkv2_m12_7 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_qkv=2,tile_mlp1=7,tile_mlp2=7
run qkv2_m12o_7 "" SVCMI_WHISPER_TUNE=tile_qkv=2,tile_mlp1=7,tile_mlp2=7,tile_o=7
run m1_3 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_mlp1=3
run all_3 "--no-single-stream" SVCMI_WHISPER_TUNE=tile_qkv=3,tile_mlp1=3,tile_mlp2=3
run base_again "" A=1
timeout 100 python scripts/microbench.py wtune > $OUT/wtune.log 2>&1; grep -E "T=500 " $OUT/wtune.log
echo "== done"
