#!/usr/bin/env python
"""With clips in flight a GEMM no longer has to fill 256 CUs on its own: sweep the Whisper encoder's tile / split-K table with 1 and 4
lanes.  Tuning aid: python scripts/lanes_tiles_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

from svcmi import Ops, weights as PW  # noqa: E402
from svcmi.lanes import GraphLanes  # noqa: E402
from svcmi.whisper.inference import WhisperEncoderModel  # noqa: E402
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
    dev = torch.device("cuda")
    ops = Ops()
    hp = C.base_hp()
    whisper = WhisperEncoderModel(None, dev, ops=ops, packed=PW.WhisperWeights(W.make_whisper_state(C.WHISPER_LARGE_V2), dev))
    enc = whisper.encoder

    def enc_fn(i):
        mel = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)["mel"].to(dev)
        return lambda: whisper.encoder(mel, torch.randn_like(mel), 0.1)[:, :500]

    # (tile_qkv, tile_o, tile_mlp1, tile_mlp2, split_o, split_mlp); tiles: 0 auto, 1 64x64, 2 128x64, 3 128x128, 6 P16 64x80
    table = [(0, 6, 6, 6, 2, 4), (1, 1, 1, 1, 2, 4), (1, 1, 1, 1, 1, 2), (1, 1, 1, 1, 2, 2), (1, 1, 1, 1, 2, 8), (1, 1, 1, 1, 4, 4), (1, 1, 1, 1, 4, 8),
             (1, 1, 1, 1, 3, 5), (1, 6, 1, 6, 2, 4), (1, 1, 1, 6, 2, 4), (1, 6, 1, 1, 2, 4), (1, 1, 6, 1, 2, 4), (1, 1, 6, 1, 4, 8)]
    if len(sys.argv) > 1:
        table = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for cfg in table:
        enc.tile_qkv, enc.tile_o, enc.tile_mlp1, enc.tile_mlp2, enc.split_o, enc.split_mlp = cfg
        try:
            r1 = rate(GraphLanes([enc_fn(0)]))
            r4 = rate(GraphLanes([enc_fn(i) for i in range(4)]))
            print(f"tiles qkv/o/mlp1/mlp2 {cfg[:4]} split o/mlp {cfg[4:]}: 1 lane {r1:.3f} ms, 4 lanes {r4:.3f} ms / clip ({472.0 / r4:.1f} TFLOP/s)", flush=True)
        except Exception as e:      # noqa: BLE001
            print(cfg, "failed:", repr(e)[:200], flush=True)


if __name__ == "__main__":
    main()
