#!/bin/bash
# Two ranks on ONE GPU (gloo for the collectives, RCCL refuses duplicate devices): exercises the N > 1 path of bench.py on hardware --
# packed-arena broadcast, per-rank lanes, barrier + max-over-ranks timing -- for configs[1] and configs[3].  Not a scaling measurement.
export SVCMI_DIST_BACKEND=gloo
for C in 1 3; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config $C \
      --steps $([ $C = 3 ] && echo 1 || echo 8) --warmup 1 --no-roofline $([ $C = 3 ] && echo "--utterances 64") 2>gpurun_out/multirank_c$C.err | tail -c 700
  echo
done
