#!/usr/bin/env python
"""svc_infer on a long clip (base.yaml, T frames) with 1 / 2 / 3 / 4 chunk streams: ms per conversion.  python scripts/chunk_streams_probe.py [T]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

from svcmi import DummyRetrieval, Ops, SynthesizerInfer, svc_infer, weights as PW  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 18000        # 3 minutes
    dev = torch.device("cuda")
    ops = Ops()
    hp = C.base_hp()
    m = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
    m.load_packed(PW.VitsWeights(W.make_vits_state(hp, seed=1234), hp, dev), dev)
    g = torch.Generator().manual_seed(0)
    ppg = torch.randn(T, hp.vits.ppg_dim, generator=g).to(dev)
    vec = torch.randn(T, hp.vits.vec_dim, generator=g).to(dev)
    pit = I.synth_f0(T, seed=3).to(dev)
    spk = I.synth_spk(hp.vits.spk_dim, seed=7)
    for prec in (None, "bf16x3"):
        m.precision = prec
        for n in (1, 2, 3, 4):
            m.chunk_streams = n
            for _ in range(2):
                svc_infer(m, DummyRetrieval(), spk, pit, ppg, vec, hp, dev, write_pit_wav=False, return_tensor=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                svc_infer(m, DummyRetrieval(), spk, pit, ppg, vec, hp, dev, write_pit_wav=False, return_tensor=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 5 * 1e3
            print(f"{prec or 'f32'} T={T} ({T / 100:.0f} s): chunk_streams {n}: {ms:.1f} ms per conversion = {T / 100 / ms * 1e3:.0f} audio-s/s", flush=True)


if __name__ == "__main__":
    main()
