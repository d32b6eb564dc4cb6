#!/bin/bash
# Round 6, session 2: the shader clock inside the GEMM K loops in the judged launch regime (4 clips in flight) and one clip at a time.
TAG=${1:-r06d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
K=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_ktrace.so
SVCMI_KTRACE_CLOCK=1 SVCMI_LIB=$K timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream > $OUT/clock_inflight.json 2> $OUT/clock_inflight.err; grep "ktrace clock" $OUT/clock_inflight.err
SVCMI_KTRACE_CLOCK=1 SVCMI_LIB=$K timeout 200 python bench.py --inflight 1 --steps 40 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/clock_single.json 2> $OUT/clock_single.err; grep "ktrace clock" $OUT/clock_single.err
timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('shipped lib', d['value'], d['ms_per_step'])"
rocm-smi --showclocks 2>/dev/null | head -20
echo "== done"
