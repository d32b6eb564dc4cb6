"""Whole conversions through ClipLanes, 4 lanes, many different requests, every reduced-precision mode: each result against the same conversion run
eagerly on the current stream, bit for bit (round 6: after the packed-fp32 operand-select hazard was removed from the kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
from svcmi.lanes import want_hw_queues
want_hw_queues(8)
import torch
from svcmi import Ops
from tests import engine_cases as E
ops = Ops()
for prec in ("f16", "mixed", "bf16x3", None):
    for T, layers in ((300, 4), (1000, 8)):
        t0 = time.time()
        try:
            E.check_clip_lanes(ops, "cuda", lanes=4, requests=int(os.environ.get("SOAK_REQUESTS", 24)), T=T, layers=layers, precision=prec)
            print(f"[lanes soak] precision {prec or 'f32'} T={T} whisper layers={layers}: 24 requests through 4 lanes bit-identical to their eager runs ({time.time() - t0:.0f} s)", flush=True)
        except AssertionError as e:
            print(f"[lanes soak] precision {prec or 'f32'} T={T}: DIFFERS: {str(e)[:300]}", flush=True)
