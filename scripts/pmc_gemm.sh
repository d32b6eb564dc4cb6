#!/bin/bash
# PMC pass over the whisper-shape GEMM microbench (counters in their own run; kernel-trace only).
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" ; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$N -o pmc -- python $ROOT/scripts/microbench.py ${2:-gemmpmc} > $OUT/$N.log 2>&1
  echo "$N rc=$?"
done
find $OUT -name "*counter_collection.csv" | head
