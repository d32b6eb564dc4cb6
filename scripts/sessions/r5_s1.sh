#!/bin/bash
# Round 5, GPU session 1: (1) the 2-deep DMA ring of the single-launch fp32 GEMMs (-DSVCMI_GEMM_NST=2) alone and with clips in flight,
# (2) lanes sweep of the judged line, (3) the three build experiments round 4 prepared (ADDR32=2, SNAKE_INTERLEAVE, both).
# Variants are built HERE first (scripts/build_variant.sh <name> <flags>); then: scripts/gpu.sh --timeout 1300 -- 'bash scripts/sessions/r5_s1.sh r05a'
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
EXP=$ROOT/whisper-vits-svc_amd/svcmi/exp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -3 | tee $OUT/rocminfo.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
echo "== judged line, default build"
timeout 300 $B > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "import json;d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]);print('default', d['value'], d['ms_per_step'], d['config'].get('single_stream'), d['roofline']['frac'])"
echo "== judged line, nst2"
SVCMI_LIB=$EXP/libsvcmi_nst2.so timeout 300 $B > $OUT/bench_nst2.json 2> $OUT/bench_nst2.err; python -c "import json;d=json.loads(open('$OUT/bench_nst2.json').read().strip().splitlines()[-1]);print('nst2', d['value'], d['ms_per_step'], d['config'].get('single_stream'), d['roofline']['frac'])"
echo "== lanes sweep"
for L in 5 6; do
  timeout 200 $B --no-roofline --no-single-stream --inflight $L > $OUT/bench_default_l$L.json 2>/dev/null; python -c "import json;d=json.loads(open('$OUT/bench_default_l$L.json').read().strip().splitlines()[-1]);print('default lanes $L', d['value'], d['ms_per_step'])"
done
for L in 5; do
  SVCMI_LIB=$EXP/libsvcmi_nst2.so timeout 200 $B --no-roofline --no-single-stream --inflight $L > $OUT/bench_nst2_l$L.json 2>/dev/null; python -c "import json;d=json.loads(open('$OUT/bench_nst2_l$L.json').read().strip().splitlines()[-1]);print('nst2 lanes $L', d['value'], d['ms_per_step'])"
done
SVCMI_TUNE=amp_mfma=0 timeout 200 $B --no-roofline > $OUT/bench_default_nomfma.json 2>/dev/null; python -c "import json;d=json.loads(open('$OUT/bench_default_nomfma.json').read().strip().splitlines()[-1]);print('default amp_mfma=0', d['value'], d['ms_per_step'], d['config'].get('single_stream'))"
echo "== lanes probe (encoder-only / synthesizer-only lanes)"
timeout 240 python scripts/lanes_probe.py > $OUT/lanes_probe_default.log 2>&1; grep lanes $OUT/lanes_probe_default.log
SVCMI_LIB=$EXP/libsvcmi_nst2.so timeout 240 python scripts/lanes_probe.py > $OUT/lanes_probe_nst2.log 2>&1; grep lanes $OUT/lanes_probe_nst2.log
echo "== Whisper GEMM shapes alone"
SVCMI_LIB=$EXP/libsvcmi_nst2.so timeout 200 python scripts/microbench.py gemm > $OUT/micro_gemm_nst2.log 2>&1; grep -E "whisper" $OUT/micro_gemm_nst2.log | head -40
echo "== prepared variants of the half-step / SnakeAlias kernels"
for V in default a2 il a2il; do
  if [ $V = default ]; then unset SVCMI_LIB; else export SVCMI_LIB=$EXP/libsvcmi_$V.so; [ -f $SVCMI_LIB ] || { echo "$V: not built"; continue; }; fi
  timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "fp16_matrix_cores or snake or amp_block" > $OUT/pytest_$V.log 2>&1; echo "$V pytest rc=$?"; tail -1 $OUT/pytest_$V.log
  timeout 300 python scripts/microbench.py amplp snake > $OUT/micro_$V.log 2>&1; echo "$V microbench rc=$?"
  grep -E "amplp .* B=1 d=1 amp_u= 1|amplp .* B=4 d=1 amp_u= 1|^snake C=(40|20|10) " $OUT/micro_$V.log | sed 's/  max diff.*//'
done
unset SVCMI_LIB
echo "== done"
