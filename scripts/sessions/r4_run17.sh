#!/bin/bash
# round-4 GPU session 17: skewed U-tile layout (CP = 12: run stride = group width mod 64) against the round-3 layout (exp/libsvcmi_noskew.so)
TAG=${1:-r04t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake_conv or snake_post or amp_block" > $OUT/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -2 $OUT/pytest_k.log
timeout 600 python scripts/microbench.py amplp > $OUT/amplp.log 2>&1; echo "amplp rc=$?"
SVCMI_LIB=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_noskew.so timeout 600 python scripts/microbench.py amplp > $OUT/amplp_noskew.log 2>&1; echo "amplp noskew rc=$?"
paste -d'|' <(grep "amplp C=10" $OUT/amplp.log | grep "amp_u= 1" | sed 's/  max diff.*//') <(grep "amplp C=10" $OUT/amplp_noskew.log | grep "amp_u= 1" | awk -F: '{print $2}' | awk '{print $1}')
cd /tmp
for L in skew noskew; do
  LIBV=""; [ $L = noskew ] && LIBV=$ROOT/whisper-vits-svc_amd/svcmi/exp/libsvcmi_noskew.so
  SVCMI_LIB=$LIBV timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_$L -o t -- python $ROOT/bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_$L.err; echo "pmc $L rc=$?"
  python - $OUT/pmc_$L <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "snake" in k:
            acc[k.split("(")[0][-60:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: int(x) for c, x in v.items()}, "conflict share", round(v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"]), 3))
PY
  find $OUT/pmc_$L -name "*.csv" -delete
done
