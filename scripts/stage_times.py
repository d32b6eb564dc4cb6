#!/usr/bin/env python
"""In-situ time of each segment of the synthesizer on the GPU box, one clip at a time and with clips in flight: the synthesizer is
captured as HIP graphs that stop after successive segments (pitch2source + prior encoder | flow | generator pre | stage 0..4 | output
layer), replayed on 1 and 4 lanes (svcmi.lanes.GraphLanes) and differenced.  Unlike a rocprofv3 trace this measures the un-instrumented
graphs, overlap included (rocprofv3 serialises the hardware queues).  Usage: python scripts/stage_times.py [precision]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

from svcmi import Ops, SynthesizerInfer, weights as PW  # noqa: E402
from svcmi.lanes import GraphLanes  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def rate(lanes, clips=32):
    for _ in range(2 * len(lanes)):
        lanes.launch()
    lanes.synchronize()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(clips):
        lanes.launch()
    lanes.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / clips * 1e3


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "f32" else None
    dev = torch.device("cuda")
    ops = Ops()
    hp = C.base_hp()
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
    model.load_packed(PW.VitsWeights(W.make_vits_state(hp, seed=1234), hp, dev), dev)
    model.precision = prec

    def syn_fn(i):
        d = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)
        vec, pit, spk, lens = d["vec"].to(dev), d["pit"].to(dev), d["spk"].to(dev), d["lengths"].to(dev, torch.int32)
        ppg50 = torch.randn(1, 500, 1280, device=dev)
        return lambda: model.inference_ppg50(ppg50, vec, pit, spk, lens, model.pitch2source(pit))

    stops = [("source+prior", "prior"), ("flow", "flow"), ("gen_pre", "gen_pre")] + [(f"stage{i}", ("stage", i)) for i in range(5)] + [("post", None)]
    prev = [0.0, 0.0]
    for name, stop in stops:
        model._stop_after = stop
        r = [rate(GraphLanes([syn_fn(i) for i in range(n)])) for n in (1, 4)]
        print(f"{prec or 'f32'} {name:13s} cumulative 1 lane {r[0]:6.3f} ms, 4 lanes {r[1]:6.3f} ms / clip   segment {r[0] - prev[0]:6.3f} / {r[1] - prev[1]:6.3f} ms",
              flush=True)
        prev = r
    model._stop_after = None


if __name__ == "__main__":
    main()
