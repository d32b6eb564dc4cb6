#!/bin/bash
# Round 6, session 9: the in-flight timeline of the GEMM family (records written by the kernels: the timeline probe build), 4 clips in flight and one at a time.
TAG=${1:-r06r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
K=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_timeline.so
SVCMI_TIMELINE=$OUT/timeline_inflight.json SVCMI_LIB=$K timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/tl_inflight_bench.json 2> $OUT/tl_inflight.err; grep "timeline:" $OUT/tl_inflight.err | cut -c1-900
SVCMI_TIMELINE=$OUT/timeline_single.json SVCMI_LIB=$K timeout 300 python bench.py --inflight 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/tl_single_bench.json 2> $OUT/tl_single.err; grep "timeline:" $OUT/tl_single.err | cut -c1-900
timeout 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('shipped lib on this box', d['value'], d['ms_per_step'], d['config']['single_stream'])"
echo "== done"
