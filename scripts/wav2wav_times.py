#!/usr/bin/env python
"""Stage times of the whole wav -> wav conversion of one 10 s clip on the GPU box (rows N2 / N3 around the judged hot path):
log-mel front-end, Whisper PPG, HuBERT-Soft units, CREPE F0 with Viterbi, then pitch2source + synthesis.  Full-size seeded
models, eager launches (no HIP graph), wall time per stage after a warm-up pass.  Usage: python scripts/wav2wav_times.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from svcmi import Ops, SynthesizerInfer  # noqa: E402
from svcmi.hubert import inference as HI  # noqa: E402
from svcmi.pitch import inference as PI  # noqa: E402
from svcmi.whisper import audio as WA  # noqa: E402
from svcmi.whisper import inference as WI  # noqa: E402
from workload import config as C  # noqa: E402
from workload import inputs as I  # noqa: E402
from workload import weights as W  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    prec = sys.argv[2] if len(sys.argv) > 2 else None      # GEMM operand precision of all four networks (None = fp32)
    ops, dev, hp = Ops(), "cuda", C.base_hp()
    n = int(16000 * secs)
    g = torch.Generator().manual_seed(0)
    t = torch.arange(n) / 16000.0
    wav = (0.4 * torch.sin(2 * np.pi * 220.0 * t * (1 + 0.05 * torch.sin(2 * np.pi * 0.7 * t))) + 0.02 * torch.randn(n, generator=g)).float()
    whisper = WI.load_model(W.make_whisper_state(C.WHISPER_LARGE_V2), dev, ops=ops)
    hubert = HI.load_model(W.make_hubert_state(C.HUBERT_SOFT), dev, ops=ops)
    crepe = PI.load_crepe(W.make_crepe_state("full"), dev, ops=ops)
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
    model.load_state_dict(W.make_vits_state(hp, seed=1234))
    model.eval()
    model.to(dev)
    whisper.encoder.precision = hubert.precision = crepe.precision = model.precision = prec
    wav_d = wav.to(dev)
    rows = []
    ms, mel = timed(lambda: WA.log_mel_spectrogram(wav_d, ops=ops, device=dev))
    rows.append(("log-mel front-end", ms))
    keep = n // 320
    ms, ppg = timed(lambda: WI.pred_ppg_from_mel(whisper, [mel], [keep]))
    rows.append(("Whisper PPG (24 blocks)", ms))
    ms, vec = timed(lambda: hubert.units(wav_d.view(1, 1, -1)))
    rows.append(("HuBERT-Soft units", ms))
    ms, pit = timed(lambda: PI.compute_f0_sing(wav, dev, model=crepe))
    rows.append(("CREPE-full F0 + Viterbi (incl. H2D / D2H)", ms))
    ppg_t = torch.repeat_interleave(ppg, 2, 0)
    vec_t = torch.repeat_interleave(vec[0], 2, 0)
    T = min(len(pit), ppg_t.shape[0], vec_t.shape[0])
    spk = I.synth_spk(hp.vits.spk_dim, seed=7).to(dev)
    # random-init CREPE weights give an arbitrary track; keep it finite and in the singing range so the generator sees normal input
    pit_t = torch.as_tensor(np.clip(np.nan_to_num(pit[:T], nan=220.0), 60.0, 900.0)).float().to(dev)

    def synth():
        src = model.pitch2source(pit_t[None])
        return model.inference(ppg_t[None, :T], vec_t[None, :T], pit_t[None], spk[None], torch.tensor([T]), src)

    ms, out = timed(synth)
    rows.append(("pitch2source + prior / flow / generator (eager)", ms))
    total = sum(r[1] for r in rows)
    for name, ms in rows:
        print(f"{name:52s} {ms:8.2f} ms")
    print(f"precision: {prec or 'fp32'}")
    print(f"{'total':52s} {total:8.2f} ms for {secs:g} s of audio = {secs * 1e3 / total:.0f}x real time (T = {T} frames, out {tuple(out.shape)})")


if __name__ == "__main__":
    main()
