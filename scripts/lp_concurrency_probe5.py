"""Two synthesizer graphs in flight with a PHASE OFFSET between them (a device-side sleep in front of the second): are the results of a
graph independent of which kernels of the other graph run beside it?  fp32 beside fp32, and amp3=f16 beside fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops
from tests import engine_cases as E
from workload import config as C, inputs as I
ops = Ops(); hp = C.base_hp(); T = int(os.environ.get("PROBE_T", 300))
F32 = "enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f32,amp4=f32,encattn=f32"
def pol(**kw):
    d = dict(item.split("=") for item in F32.split(",")); d.update(kw)
    return "mixed:" + ",".join(f"{k}={v}" for k, v in d.items())
ms = [E.make_model(hp, ops, "cuda")[0] for _ in range(2)]
ds = [{k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=80 + i, B=1).items()} for i in range(2)]
lens = [torch.tensor([T], dtype=torch.int32, device="cuda") for _ in range(2)]
def build(m, d, ln):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    def fn():
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        return m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], ln, src, noise=d["enc_noise"])
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
for pa in (None, pol(amp3="f16")):
    ms[0].precision = pa
    gs = [build(ms[0], ds[0], lens[0]), build(ms[1], ds[1], lens[1])]
    torch.cuda.synchronize()
    ref = []
    for g, o, s in gs:
        with torch.cuda.stream(s):
            g.replay()
        s.synchronize(); ref.append(o.clone())
    for first in (0, 1):
        for off in (0, 100_000, 300_000, 600_000, 1_000_000, 1_500_000, 2_000_000, 3_000_000, 4_000_000, 6_000_000):
            worst = [0.0, 0.0]
            for rep in range(4):
                order = [gs[first], gs[1 - first]]
                with torch.cuda.stream(order[0][2]):
                    order[0][0].replay()
                with torch.cuda.stream(order[1][2]):
                    torch.cuda._sleep(off)
                    order[1][0].replay()
                torch.cuda.synchronize()
                e = [float((o - r).abs().max()) for (g, o, s), r in zip(gs, ref)]
                worst = [max(a, b) for a, b in zip(worst, e)]
            if max(worst) > 0:
                print(f"[probe5] A = {'f32' if pa is None else 'amp3=f16'}, B = f32; {'A' if first == 0 else 'B'} first, the other {off} sleep cycles later: worst A {worst[0]:.3e} B {worst[1]:.3e}", flush=True)
    print(f"[probe5] A = {'f32' if pa is None else 'amp3=f16'} done", flush=True)
