#!/bin/bash
# Round 5, GPU session 17: the Whisper window GEMMs one by one, mid-barrier loop (default build) vs -DSVCMI_GEMM_MIDBAR=0
TAG=${1:-r05r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
for V in default nomidbar default nomidbar; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$EXP/libsvcmi_$V.so; fi
  timeout 300 python scripts/microbench.py wtune > $OUT/micro_$V.log 2>&1
  echo "--- $V"; grep -E "T=500 .*(whisper_qkv .*tile=1 split=1|whisper_mlp1 .*tile=6 split=1|whisper_o .*tile=6 split=2|whisper_mlp2 .*tile=6 split=4)" $OUT/micro_$V.log
done
unset SVCMI_LIB
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $ROOT/scripts/microbench.py wtune > /dev/null 2>&1; cd $ROOT
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seen = {}
for r in rows:
    k = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    seen.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(seen.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(k, len(v), round(sorted(v)[len(v) // 2], 2))
PY
find $OUT/prof -name "*.csv" -delete
echo "== done"
