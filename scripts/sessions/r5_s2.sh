#!/bin/bash
# Round 5, GPU session 2: which launches should take the low-LDS forms when 4 clips share the chip?  The judged line (configs[1], fp32,
# 4 clips in flight) under (1) ring2 masks of the single-launch GEMMs (bit 0 qkv, 1 out-proj, 2 mlp-up, 3 mlp-down, 4 synthesizer),
# (2) ring depth of the grouped generator GEMMs, (3) the U-tile (LDS-heavy) vs the recompute form of the narrow half-steps.
# scripts/gpu.sh --timeout 900 -- 'bash scripts/sessions/r5_s2.sh r05b'
TAG=${1:-r05b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'])" 2>/dev/null || echo "$2 FAILED"; }
run() {  # name, env assignments...
  local name=$1; shift
  env "$@" timeout 200 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; show $OUT/bench_$name.json "$name"
}
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "two_deep or snake_conv_group or fp16_matrix" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
for M in 0 31 15 16 5 10 12 3 21 26; do run ring$M SVCMI_RING2=$M; done
run ring0_again SVCMI_RING2=0
run ring31_gnst2 SVCMI_RING2=31 SVCMI_TUNE=group_nst=2
run ring31_gnst3 SVCMI_RING2=31 SVCMI_TUNE=group_nst=3
run ring31_noU SVCMI_RING2=31 SVCMI_TUNE=amp_u=-1
run ring31_nomfma SVCMI_RING2=31 SVCMI_TUNE=amp_mfma=0
echo "== full default line (roofline + single stream)"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err; python -c "import json;d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]);print('full', d['value'], d['ms_per_step'], d['config'].get('single_stream'), d['roofline']['frac'], d['config']['gemm_ring2_mask'])"
echo "== done"
