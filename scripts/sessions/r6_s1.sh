#!/bin/bash
# Round 6, session 1: what bounds a K-step of the M = 500 GEMMs?  The shipped library against four timing-probe builds of the same K loop
# (no DMA / no MFMA / fill only / MFMA only): scripts/microbench.py kprobe.
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
for v in base nodma nomfma fillonly nolds; do
  echo "== $v"
  if [ $v = base ]; then timeout 300 python scripts/microbench.py kprobe 2>&1 | grep -v amdgpu.ids | tee $OUT/kprobe_$v.log
  else SVCMI_LIB=$EXP/libsvcmi_$v.so timeout 300 python scripts/microbench.py kprobe 2>&1 | grep -v amdgpu.ids | tee $OUT/kprobe_$v.log; fi
done
echo "== done"
