#!/usr/bin/env python
"""Where does a 4-lane step spend its time?  Times the two halves of the judged batch-1 step separately -- the Whisper encoder
(472 GFLOP of GEMMs + attention) and the synthesizer (prior / flow / generator) -- with 1 / 2 / 4 clips in flight each, and the
mixture (half the lanes encoding, half synthesising).  Tuning aid: python scripts/lanes_probe.py [precision]"""
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
from svcmi.whisper.inference import WhisperEncoderModel  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def rate(lanes, clips=40):
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
    whisper = WhisperEncoderModel(None, dev, ops=ops, packed=PW.WhisperWeights(W.make_whisper_state(C.WHISPER_LARGE_V2), dev))
    model.precision = whisper.encoder.precision = prec

    def enc_fn(i):
        d = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)
        mel = d["mel"].to(dev)
        return lambda: whisper.encoder(mel, torch.randn_like(mel), 0.1)[:, :500]

    def syn_fn(i):
        d = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)
        vec, pit, spk, lens = d["vec"].to(dev), d["pit"].to(dev), d["spk"].to(dev), d["lengths"].to(dev, torch.int32)
        ppg50 = torch.randn(1, 500, 1280, device=dev)
        return lambda: model.inference_ppg50(ppg50, vec, pit, spk, lens, model.pitch2source(pit))

    for n in (1, 2, 4):
        e = rate(GraphLanes([enc_fn(i) for i in range(n)]))
        s = rate(GraphLanes([syn_fn(i) for i in range(n)]))
        print(f"{prec or 'f32'}: {n} lanes: encoder {e:.3f} ms / clip ({472.0 / e:.1f} TFLOP/s of GEMMs), synthesizer {s:.3f} ms / clip; sum {e + s:.3f}", flush=True)
    for ne, ns in ((1, 1), (2, 2), (1, 3), (3, 1)):
        fns = [enc_fn(i) for i in range(ne)] + [syn_fn(i) for i in range(ns)]
        # interleave so that round-robin alternates the kinds
        order = []
        for k in range(max(ne, ns)):
            if k < ne:
                order.append(fns[k])
            if k < ns:
                order.append(fns[ne + k])
        m = rate(GraphLanes(order), clips=40)
        print(f"{prec or 'f32'}: {ne} encoder + {ns} synthesizer lanes, round-robin: {m:.3f} ms per launched half-clip", flush=True)


if __name__ == "__main__":
    main()
