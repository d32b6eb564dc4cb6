#!/bin/bash
# round 2, run 11: staged lanes (encoder tokens) sweep + tests.  Record of a rejected experiment: the --encoder-tokens flag (Whisper and
# synthesizer captured as two stages, at most N clips inside the Whisper stage) was removed afterwards -- profiles/r02q_staged_lanes_experiment.log
mkdir -p gpurun_out/r2q
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -k "ivf or clips_in_flight" 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline --no-roofline --steps 48 "$@" 2>gpurun_out/r2q/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('single_stream'))" || tail -5 gpurun_out/r2q/err.log; }
for tok in 0 1 2; do for n in 2 3 4 6; do echo "tokens $tok inflight $n"; b --inflight $n --encoder-tokens $tok; done; done
for p in bf16x3 f16; do for tok in 1 2; do echo "$p tokens $tok inflight 4"; b --precision $p --inflight 4 --encoder-tokens $tok; done; done
