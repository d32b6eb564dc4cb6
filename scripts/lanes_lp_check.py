"""Is a reduced-precision conversion the same bits through ClipLanes (HIP graph replay on a lane's stream) as run eagerly?  (round 6, r06x)
Usage: python scripts/lanes_lp_check.py [precision]   -- prints max |graph - eager| per variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops
from svcmi.serving import ClipLanes, convert_step
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
d = {k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=70, B=1, ppg=False).items()}
lens = torch.tensor([T], dtype=torch.int32, device="cuda")
noise = {k: d[k] for k in ("mel_noise", "rand_ini", "src_noise", "enc_noise")}
buf = dict(mel=d["mel"], vec=d["vec"], pit=d["pit"], spk=d["spk"], lengths=lens)
e1 = convert_step(m, wm, buf, T // 2, noise).clone()
e2 = convert_step(m, wm, buf, T // 2, noise).clone()
print(f"[{prec}] eager vs eager: {float((e1 - e2).abs().max()):.3e}")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    e3 = convert_step(m, wm, buf, T // 2, noise).clone()
s.synchronize()
print(f"[{prec}] eager on another stream vs eager: {float((e1 - e3).abs().max()):.3e}")
for lanes, ring2 in ((1, 0), (2, 0), (2, 12)):
    cl = ClipLanes(m, wm, T, B=1, lanes=lanes, device="cuda", pinned_noise=True, ring2=ring2)
    outs = []
    for _ in range(lanes + 1):
        t = cl.submit(noise=noise, lengths=lens, **{k: buf[k] for k in ("mel", "vec", "pit", "spk")})
        outs.append(cl.result(t))
    print(f"[{prec}] lanes={lanes} ring2={ring2}: " + " ".join(f"{float((o - e1).abs().max()):.3e}" for o in outs))
# per stage: the Whisper encoder alone, eager vs captured
g = torch.cuda.CUDAGraph()
mel, mn = buf["mel"].clone(), noise["mel_noise"].clone()
ppg_e = wm.encoder(mel, mn, 0.1).clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    wm.encoder(mel, mn, 0.1)
    s.synchronize()
    with torch.cuda.graph(g, stream=s):
        ppg_g = wm.encoder(mel, mn, 0.1)
g.replay()
torch.cuda.synchronize()
print(f"[{prec}] whisper encoder graph vs eager: {float((ppg_g - ppg_e).abs().max()):.3e}")
