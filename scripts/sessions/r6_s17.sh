#!/bin/bash
# Round 6, session 17: the narrow vector half-step with its weights as a pair-ordered LDS image: parity (GPU tests of the touched kernels), launch times
# alone and in the pipeline, hoisted weight reads (131 VGPRs) against one input slice at a time (-DSVCMI_AMP_WL_BARRIER, 114 VGPRs), the judged line
TAG=${1:-r06zd}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "snake_conv or in_flight_beside" > $OUT/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"; tail -2 $OUT/pytest_kernels.log
for V in old new wlbar; do L=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so; [ $V != new ] && L=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so; echo "== microbench $V"; SVCMI_LIB=$L python scripts/microbench.py ampgroup 2>&1 | grep -E "C=10.*(B=1|B=4) amp_u=1|C=20.*amp_u"; done
for V in new wlbar old; do
  L=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so; [ $V != new ] && L=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so
  cd /tmp; SVCMI_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$V -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_${V}_bench.json 2> $OUT/prof_$V.err
  cd $ROOT; python scripts/prof_summary.py $OUT/prof_$V $OUT/kernel_stats_$V.csv 11 > /dev/null 2>&1; echo "== pipeline $V"; grep -E "snake_conv_group_u_kernel<12,10,1,1>,[0-9]+,|total kernel" $OUT/kernel_stats_$V.csv | cut -c1-160
  rm -rf $OUT/prof_$V
done
for V in new wlbar old new wlbar old; do L=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so; [ $V != new ] && L=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_$V.so
  SVCMI_LIB=$L timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('judged line $V', d['value'], d['ms_per_step'], 'single', d['config'].get('single_stream',{}).get('ms_per_step'))"; done
