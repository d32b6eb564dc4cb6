#!/bin/bash
# round 2, run 8: IVF-Flat kernels on the GPU + clips-in-flight experiment
set -x
mkdir -p gpurun_out/r2n
python -m pytest tests/test_gpu_kernels.py -x -q -k "ivf or knn" 2>&1 | tail -5
python -m pytest tests/test_abi.py -x -q 2>&1 | tail -2
for n in 1 2 3; do
  python bench.py --no-cpu-baseline --no-roofline --inflight $n --steps 40 > gpurun_out/r2n/bench_inflight$n.json 2> gpurun_out/r2n/bench_inflight$n.log; tail -c 600 gpurun_out/r2n/bench_inflight$n.json
done
python bench.py --no-cpu-baseline --no-roofline --inflight 2 --precision bf16x3 --steps 40 > gpurun_out/r2n/bench_inflight2_bf16x3.json 2>/dev/null; tail -c 400 gpurun_out/r2n/bench_inflight2_bf16x3.json
python bench.py --no-cpu-baseline --no-roofline --inflight 2 --precision f16 --steps 40 > gpurun_out/r2n/bench_inflight2_f16.json 2>/dev/null; tail -c 400 gpurun_out/r2n/bench_inflight2_f16.json
python bench.py --config 2 --no-roofline --inflight 2 > gpurun_out/r2n/bench_c2_inflight2.json 2>/dev/null; tail -c 400 gpurun_out/r2n/bench_c2_inflight2.json
python bench.py --config 4 --no-roofline --inflight 2 > gpurun_out/r2n/bench_c4_inflight2.json 2>/dev/null; tail -c 400 gpurun_out/r2n/bench_c4_inflight2.json
