#!/bin/bash
# round 2, run 9: IVF tests again + clips-in-flight sweep (streams / HW queues)
mkdir -p gpurun_out/r2o
python -m pytest tests/test_gpu_kernels.py -x -q -k "ivf" 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline --no-roofline --steps 48 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'])"; }
for n in 2 3 4 6 8; do echo "inflight $n"; b --inflight $n; done
export GPU_MAX_HW_QUEUES=8
for n in 2 3 4 8; do echo "hwq8 inflight $n"; b --inflight $n; done
export GPU_MAX_HW_QUEUES=2
for n in 2 3 4; do echo "hwq2 inflight $n"; b --inflight $n; done
unset GPU_MAX_HW_QUEUES
echo "c2 inflight 3"; b --config 2 --inflight 3
echo "c4 inflight 3"; b --config 4 --inflight 3
echo "f16 inflight 3"; b --precision f16 --inflight 3
echo "bf16x3 inflight 3"; b --precision bf16x3 --inflight 3
