"""Which neighbour disturbs a reduced-precision synthesizer graph: graph A (policy pa) beside graph B (policy pb), different inputs."""
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
def build(m, d):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    def fn():
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        return m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
def spin_graph():          # a neighbour that only burns time on the vector ALUs / HBM (torch kernels), no svcmi kernel
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    x = torch.randn(1 << 24, device="cuda")
    def fn():
        y = x
        for _ in range(40):
            y = torch.sin(y) * 1.0001 + 0.1
        return y
    with torch.cuda.stream(s):
        fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
for pa, pb in ((None, pol(amp3="f16")), (pol(amp3="f16"), None), ("spin", pol(amp3="f16")), ("spin", pol(amp2="f16")), (None, pol(amp2="f16")), (pol(amp2="f16"), pol(amp2="f16"))):
    gs = []
    for m, d, p in zip(ms, ds, (pa, pb)):
        if p == "spin":
            gs.append(spin_graph())
        else:
            m.precision = p
            gs.append(build(m, d))
    torch.cuda.synchronize()
    ref = []
    for g, o, s in gs:
        with torch.cuda.stream(s):
            g.replay()
        s.synchronize(); ref.append(o.clone())
    worst = [0.0, 0.0]; bad = 0
    for rep in range(20):
        for g, o, s in gs:
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = [float((o - r).abs().max()) for (g, o, s), r in zip(gs, ref)]
        bad += any(x > 0 for x in e); worst = [max(a, b) for a, b in zip(worst, e)]
    short = lambda p: "f32" if p is None else (p if p == "spin" else ",".join(x for x in p[6:].split(",") if not x.endswith("f32")))
    print(f"[probe2] A = {short(pa)}  B = {short(pb)}: {bad}/20 differ, worst {worst}", flush=True)
