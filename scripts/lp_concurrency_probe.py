"""Two HIP graphs of the synthesizer in flight at once, different inputs, per precision policy: does a graph's output change when the other
graph runs beside it?  (round 6: reduced-precision ClipLanes results were not bit-identical with two lanes busy.)
Usage: python scripts/lp_concurrency_probe.py [tune=val,...] -- policies are listed below."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops
from tests import engine_cases as E
from workload import config as C, inputs as I

ops = Ops(); hp = C.base_hp(); T = int(os.environ.get("PROBE_T", 300))
m, _ = E.make_model(hp, ops, "cuda")
F32 = "enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f32,amp4=f32,encattn=f32"
def pol(**kw):
    d = dict(item.split("=") for item in F32.split(","))
    d.update(kw)
    return "mixed:" + ",".join(f"{k}={v}" for k, v in d.items())
POLICIES = [("all f32", None), ("ups f16", pol(ups="f16")), ("amp0 f16", pol(amp0="f16")), ("amp1 f16", pol(amp1="f16")), ("amp2 f16", pol(amp2="f16")),
            ("amp3 f16", pol(amp3="f16")), ("amp4 f16", pol(amp4="f16")), ("amp0 bf16x3", pol(amp0="bf16x3")), ("ups bf16x3", pol(ups="bf16x3")),
            ("enc+flow f16", pol(enc="f16", flow="f16"))]
if len(sys.argv) > 1:
    POLICIES = [p for p in POLICIES if p[0] in sys.argv[1:]] or POLICIES
ds = [{k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=80 + i, B=1).items()} for i in range(2)]
lens = torch.tensor([T], dtype=torch.int32, device="cuda")
for name, p in POLICIES:
    m.precision = p
    graphs, outs, streams = [], [], []
    for d in ds:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        def fn(d=d):
            src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
            return m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])
        with torch.cuda.stream(s):
            fn(); fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o = fn()
        graphs.append(g); outs.append(o); streams.append(s)
    torch.cuda.synchronize()
    ref = []
    for g, o, s in zip(graphs, outs, streams):        # one at a time
        with torch.cuda.stream(s):
            g.replay()
        s.synchronize(); ref.append(o.clone())
    worst = [0.0, 0.0]; bad = 0
    for rep in range(20):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = [float((o - r).abs().max()) for o, r in zip(outs, ref)]
        bad += any(x > 0 for x in e); worst = [max(a, b) for a, b in zip(worst, e)]
    print(f"[probe T={T}] {name}: {bad}/20 concurrent replays differ, worst {worst}", flush=True)
