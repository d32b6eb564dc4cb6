#!/bin/bash
# generator-stage stream scheduling experiments (bench.py graph replay, no CPU baseline)
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $1"; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run base SVCMI_AMP_ORDER=asc
run desc SVCMI_AMP_ORDER=desc
run base_q8 SVCMI_AMP_ORDER=asc GPU_MAX_HW_QUEUES=8
run desc_q8 SVCMI_AMP_ORDER=desc GPU_MAX_HW_QUEUES=8
run base2 SVCMI_AMP_ORDER=asc
