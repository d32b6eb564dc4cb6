"""Does a reduced-precision conversion depend on what the PREVIOUS conversion left in the workspace?  A, B, A again: the two A results must be
the same bits (round 6, r06y: tests/test_gpu_engine.py::test_clip_lanes_capture_packs_the_16bit_weight_images found 1.4e-4 between a lane's
replay and the eager run of the same request).  Usage: python scripts/lp_state_check.py [precision]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops
from svcmi.whisper.inference import load_model
from tests import engine_cases as E
from workload import config as C, inputs as I, weights as W

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
ops = Ops()
hp = C.base_hp()
T = 300
m, _ = E.make_model(hp, ops, "cuda")
wm = load_model(W.make_whisper_state(dict(C.WHISPER_LARGE_V2, n_audio_layer=4)), "cuda", ops=ops)
m.precision = prec
wm.encoder.precision = "f16" if prec.startswith("mixed") else prec
reqs = []
for seed in (70, 71):
    d = {k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=seed, B=1, ppg=False).items()}
    reqs.append(d)
lens = torch.tensor([T], dtype=torch.int32, device="cuda")


def whisper(d):
    return wm.encoder(d["mel"], d["mel_noise"], 0.1)[:, :T // 2].clone()


def synth(d, ppg50):
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    ppg = torch.repeat_interleave(ppg50, 2, dim=1)[:, :T].contiguous()
    wave, parts = m.inference(ppg, d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"], return_parts=True)
    return dict(source=src.clone(), z_p=parts["z_p"].clone(), z=parts["z"].clone(), wave=wave.clone())


A, B = reqs
pa1 = whisper(A); whisper(B); pa2 = whisper(A)
print(f"[{prec}] whisper A,B,A: {float((pa1 - pa2).abs().max()):.3e}")
pb = whisper(B)
s1 = synth(A, pa1); synth(B, pb); s2 = synth(A, pa1)
for k in s1:
    print(f"[{prec}] synth {k}: A,B,A {float((s1[k] - s2[k]).abs().max()):.3e}  (|.| max {float(s1[k].abs().max()):.3f})")
# the same on a workspace that held something else entirely in between: a different clip LENGTH
d3 = {k: v.to("cuda") for k, v in I.synth_clip(T=200, hp=hp, seed=72, B=1, ppg=False).items()}
p3 = wm.encoder(d3["mel"], d3["mel_noise"], 0.1)[:, :100].clone()
src3 = m.pitch2source(d3["pit"], noise=(d3["rand_ini"], d3["src_noise"]))
m.inference(torch.repeat_interleave(p3, 2, dim=1)[:, :200].contiguous(), d3["vec"], d3["pit"], d3["spk"], torch.tensor([200], dtype=torch.int32, device="cuda"), src3, noise=d3["enc_noise"])
pa3 = whisper(A)
s3 = synth(A, pa1)
print(f"[{prec}] whisper A after a T = 200 clip: {float((pa1 - pa3).abs().max()):.3e}")
for k in s1:
    print(f"[{prec}] synth {k}: A after a T = 200 clip {float((s1[k] - s3[k]).abs().max()):.3e}")
