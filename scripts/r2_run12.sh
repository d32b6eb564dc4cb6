#!/bin/bash
# round 2, run 12: lanes confined to CU partitions (hipExtStreamCreateWithCUMask).  Record of a rejected experiment: the --cu-partition flag and
# the svcmi_stream_create_cu_mask entry point were removed afterwards -- profiles/r02r_cu_partition_experiment.log
mkdir -p gpurun_out/r2r
python -m pytest tests/test_gpu_kernels.py tests/test_abi.py -x -q -k "ivf or abi" 2>&1 | tail -3
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 48 "$@" 2>gpurun_out/r2r/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('single_stream'))" || tail -5 gpurun_out/r2r/err.log; }
for sch in xcd spread; do for n in 2 4 8; do echo "$sch inflight $n"; b --inflight $n --cu-partition $sch; done; done
for p in bf16x3 f16; do for sch in xcd spread; do echo "$p $sch inflight 4"; b --precision $p --inflight 4 --cu-partition $sch; done; done
echo "c4 xcd 4"; b --config 4 --steps 24 --inflight 4 --cu-partition xcd
echo "c2 xcd 2"; b --config 2 --steps 12 --inflight 2 --cu-partition xcd
