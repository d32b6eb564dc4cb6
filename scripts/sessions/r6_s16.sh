#!/bin/bash
# Round 6, session 16: A/B in the pipeline (rocprofv3, one clip at a time) of the library before the operand-select rewrite (SVCMI_LIB=exp/libsvcmi_old.so) and after
TAG=${1:-r06zc}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in new old; do
  L=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so; [ $V = old ] && L=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_old.so
  cd /tmp; SVCMI_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$V -o trace -- python $ROOT/bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_${V}_bench.json 2> $OUT/prof_$V.err
  cd $ROOT; python scripts/prof_summary.py $OUT/prof_$V $OUT/kernel_stats_$V.csv 11 > /dev/null 2>&1; echo "== $V"; grep -E "snake_conv_group_u|snake_convm|snake_alias|upsample_noise|total kernel" $OUT/kernel_stats_$V.csv | cut -c1-160
  rm -rf $OUT/prof_$V
done
for V in old new; do L=$ROOT/whisper-vits-svc_amd/svcmi/libsvcmi.so; [ $V = old ] && L=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_old.so; echo "== microbench $V"; SVCMI_LIB=$L python scripts/microbench.py ampgroup 2>&1 | grep "C=10.*B=1 amp_u=1"; done
