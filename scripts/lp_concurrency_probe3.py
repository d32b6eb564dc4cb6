"""Address ranges of everything two synthesizer graphs (A reduced precision, built first; B fp32) touch: do any overlap?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops
from tests import engine_cases as E
from workload import config as C, inputs as I
ops = Ops(); hp = C.base_hp(); T = 300
F32 = "enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f32,amp4=f32,encattn=f32"
def pol(**kw):
    d = dict(item.split("=") for item in F32.split(",")); d.update(kw)
    return "mixed:" + ",".join(f"{k}={v}" for k, v in d.items())
ms = [E.make_model(hp, ops, "cuda")[0] for _ in range(2)]
ds = [{k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=80 + i, B=1).items()} for i in range(2)]
lens = torch.tensor([T], dtype=torch.int32, device="cuda")
def build(m, d, parts, stop=None):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    def fn():
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        if stop is not None:
            m._stop_after = stop
            ppg50 = d["ppg"][:, ::2].contiguous()
            o = m.inference_ppg50(ppg50, d["vec"], d["pit"], d["spk"], lens, src.view(1, 1, -1), noise=d["enc_noise"])
            m._stop_after = None
            return [o]
        if parts:
            w, pr = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"], return_parts=True)
            return [w, pr["z_p"], pr["z"], src]
        return [m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])]
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
ms[0].precision = pol(amp3="f16")
def run(gs, tag):
    torch.cuda.synchronize()
    ref = []
    for g, o, s in gs:
        with torch.cuda.stream(s):
            g.replay()
        s.synchronize(); ref.append([x.clone() for x in o])
    worst = None
    for rep in range(20):
        for g, o, s in gs:
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = [[float((x - r).abs().max()) for x, r in zip(o, rr)] for (g, o, s), rr in zip(gs, ref)]
        worst = e if worst is None else [[max(a, b) for a, b in zip(x, y)] for x, y in zip(worst, e)]
    print(f"[probe3] {tag}: worst per output A {worst[0]}  B {worst[1]}", flush=True)
A = build(ms[0], ds[0], False)
run([A, build(ms[1], ds[1], True)], "A amp3=f16 beside B f32 (wave, z_p, z, source)")
for stop in ("gen_pre", ("stage", 0), ("stage", 1), ("stage", 2), ("stage", 3), ("stage", 4)):
    run([A, build(ms[1], ds[1], False, stop)], f"A amp3=f16 beside B f32 truncated after {stop} (output = the wave buffer, untouched when truncated)")
