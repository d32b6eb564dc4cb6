#!/usr/bin/env python
"""The drop-in path at full model sizes, models resident (what `python -m svcmi.svc_inference` / svc_inference_batch do per file after
loading): (a) svc_infer from pre-extracted features on the HOST (numpy, as --ppg / --vec / --pit deliver them), (b) wav -> wav: the three
extractors in flight (svcmi.svc_inference.extract_features, CREPE in bf16x3 like the CLI default) + svc_infer.  Eager: no HIP graph, the
stage-level C++ host of libsvcmi.so makes the launches.  Usage: python scripts/dropin_times.py [seconds] [f0_precision]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from svcmi import Ops, SynthesizerInfer  # noqa: E402
from svcmi.hubert import inference as HI  # noqa: E402
from svcmi.pitch import inference as PI  # noqa: E402
from svcmi.svc_inference import DummyRetrieval, extract_features, svc_infer  # noqa: E402
from svcmi.whisper import inference as WI  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def main():
    print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    f0_prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    ops, dev, hp = Ops(), "cuda", C.base_hp()
    n = int(16000 * secs)
    g = torch.Generator().manual_seed(0)
    t = torch.arange(n) / 16000.0
    wav = (0.4 * torch.sin(2 * np.pi * 220.0 * t * (1 + 0.05 * torch.sin(2 * np.pi * 0.7 * t))) + 0.02 * torch.randn(n, generator=g)).float().numpy()
    whisper = WI.load_model(W.make_whisper_state(C.WHISPER_LARGE_V2), dev, ops=ops)
    hubert = HI.load_model(W.make_hubert_state(C.HUBERT_SOFT), dev, ops=ops)
    crepe = PI.load_crepe(W.make_crepe_state("full"), dev, ops=ops)
    crepe.precision = None if f0_prec == "f32" else f0_prec
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
    model.load_state_dict(W.make_vits_state(hp, seed=1234))
    model.eval()
    model.to(dev)
    model.warm()
    spk = I.synth_spk(hp.vits.spk_dim, seed=7)
    T = int(secs * 100)
    # (a) pre-extracted features on the host, as np.load / the pitch CSV deliver them
    d = I.synth_clip(T=T, hp=hp, seed=3, B=1)
    ppg_h, vec_h, pit_h = d["ppg"][0].clone(), d["vec"][0].clone(), d["pit"][0].clone()
    ms_a, out = timed(lambda: svc_infer(model, DummyRetrieval(), spk, pit_h, ppg_h, vec_h, hp, dev, write_pit_wav=False))
    print(f"svc_infer from host features (--ppg/--vec/--pit): {ms_a:7.2f} ms per {secs:g} s clip = {secs * 1e3 / ms_a:6.0f}x real time "
          f"(out {out.shape}, incl. H2D of the features and D2H of the waveform)")

    # (b) wav -> wav
    def wav2wav(in_flight=True):
        ppg, vec, f0 = extract_features(wav, whisper, hubert, crepe, dev, in_flight=in_flight)
        ppg, vec = torch.repeat_interleave(ppg, 2, 0), torch.repeat_interleave(vec, 2, 0)
        pit = torch.as_tensor(np.clip(np.nan_to_num(PI.quantize_pitch_like_csv(f0), nan=220.0), 60.0, 900.0)).float()   # random-init CREPE: keep the track sane
        return svc_infer(model, DummyRetrieval(), spk, pit, ppg, vec, hp, dev, write_pit_wav=False)
    ms_s, _ = timed(lambda: wav2wav(False))
    print(f"wav -> wav, extractors one after the other: {ms_s:7.2f} ms")
    wav_d = torch.from_numpy(wav).to(dev)
    for name, fn in (("ppg", lambda: WI.ppg_from_audio(whisper, wav)), ("vec", lambda: HI.units_windowed(hubert, wav)),
                     ("vec from a device tensor", lambda: hubert.units(wav_d.view(1, 1, -1))),
                     ("svc_infer again", lambda: svc_infer(model, DummyRetrieval(), spk, pit_h, ppg_h, vec_h, hp, dev, write_pit_wav=False)),
                     ("f0", lambda: PI.compute_f0_sing(wav, dev, model=crepe))):
        ms, _ = timed(fn)
        print(f"   {name}: {ms:6.2f} ms")
    for nb in (512, 1024, 2048):
        PI.NET_BATCH = nb
        ms, _ = timed(lambda: PI.compute_f0_sing(wav, dev, model=crepe))
        print(f"   f0 with network batches of {nb} frames: {ms:6.2f} ms")
    PI.NET_BATCH = int(os.environ.get("SVCMI_F0_BATCH", PI.NET_BATCH))
    ms_b, out = timed(wav2wav)
    print(f"wav -> wav (3 extractors in flight, CREPE {f0_prec}, fp32 elsewhere): {ms_b:7.2f} ms per {secs:g} s clip = {secs * 1e3 / ms_b:6.0f}x real time "
          f"(out {out.shape})")


if __name__ == "__main__":
    main()
