#!/bin/bash
# round 2, run 10: IVF + lanes tests, the judged line with clips in flight, sweeps
mkdir -p gpurun_out/r2p
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -k "ivf or clips_in_flight" 2>&1 | tail -3
python bench.py > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.log; tail -c 2500 gpurun_out/r2p/bench.json
b() { python bench.py --no-cpu-baseline --no-roofline --steps 48 "$@" 2>gpurun_out/r2p/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('single_stream'))" || tail -5 gpurun_out/r2p/err.log; }
for n in 3 4 5 6; do echo "inflight $n"; b --inflight $n; done
export GPU_MAX_HW_QUEUES=16
for n in 4 6 8; do echo "hwq16 inflight $n"; b --inflight $n; done
unset GPU_MAX_HW_QUEUES
for p in bf16x3 f16; do echo "$p inflight 4"; b --precision $p --inflight 4; done
for n in 2 3; do echo "c2 inflight $n"; b --config 2 --steps 12 --inflight $n; done
for n in 2 4; do echo "c4 inflight $n"; b --config 4 --steps 24 --inflight $n; done
