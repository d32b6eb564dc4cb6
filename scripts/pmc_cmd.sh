#!/bin/bash
# MFMA / LDS / wave-state counters per kernel of an arbitrary command (two rocprofv3 --pmc passes, --kernel-trace only), summarised by
# scripts/pmc_summary.py.  Usage: scripts/pmc_cmd.sh <tag> <command ...>  ->  gpurun_out/<tag>/pmc.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $ROOT && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- "$@" > $OUT/pass$i.log 2>&1)
  echo "pass$i rc=$?"
done
cd $ROOT
python scripts/pmc_summary.py $OUT $OUT/pmc.json > /dev/null 2>&1
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
python -c "
import json,sys
d=json.load(open('$OUT/pmc.json'))
for k,v in d.get('kernels', d).items():
    if isinstance(v, dict) and 'mfma_pipe_utilisation' in v: print(k, json.dumps(v))
"
