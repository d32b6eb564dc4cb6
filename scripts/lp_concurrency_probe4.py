"""Victim = ONE fp32 kernel repeated inside a HIP graph, culprit = the synthesizer with its 20-channel stage on the fp16 matrix cores, in flight
together: which kernel's results change?"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops, weights as PW
from tests import engine_cases as E
from workload import config as C, inputs as I, weights as W
ops = Ops(); hp = C.base_hp(); T = 300
F32 = "enc=f32,flow=f32,ups=f32,amp0=f32,amp1=f32,amp2=f32,amp3=f32,amp4=f32,encattn=f32"
def pol(**kw):
    d = dict(item.split("=") for item in F32.split(",")); d.update(kw)
    return "mixed:" + ",".join(f"{k}={v}" for k, v in d.items())
culprit_policy = pol(**dict(a.split("=") for a in sys.argv[1:])) if len(sys.argv) > 1 else pol(amp3="f16")
m = E.make_model(hp, ops, "cuda")[0]
m.precision = culprit_policy
d = {k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=80, B=1).items()}
lens = torch.tensor([T], dtype=torch.int32, device="cuda")
def graph(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
def culprit():
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    return [m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])]
A = graph(culprit)
gen = torch.Generator().manual_seed(5)
filt = W.kaiser_sinc_filter().view(-1).to("cuda")
def amp_problems(c, ld, n):
    probs = []
    for k, dil in ((3, 1), (11, 5), (7, 3)):
        x = torch.zeros(1, n, ld); x[..., :c] = torch.randn(1, n, c, generator=gen)
        res = torch.zeros(1, n, ld); res[..., :c] = torch.randn(1, n, c, generator=gen)
        al, be = torch.zeros(ld), torch.zeros(ld)
        al[:c], be[:c] = torch.randn(c, generator=gen) * 0.3, torch.randn(c, generator=gen) * 0.3
        w = PW.pack_conv(torch.randn(c, c, k, generator=gen) / math.sqrt(c * k), ld, ld).to("cuda")
        bias = PW.pad_vec(torch.randn(c, generator=gen), ld).to("cuda")
        probs.append(dict(x=x.to("cuda"), alpha_log=al.to("cuda"), beta_log=be.to("cuda"), w=w, bias=bias, ksize=k, dilation=dil, res=res.to("cuda"), alpha=0.5))
    return probs
R = int(os.environ.get("PROBE_R", 60))
def victim_amp(c, ld, n, knob=None):
    probs = amp_problems(c, ld, n)
    def fn():
        outs = []
        for r in range(R):
            ps = [dict(p, out=torch.empty_like(p["x"])) for p in probs]
            outs += ops.snake_conv_group(ps, filt, c=c)
        return outs
    return fn
def victim_alias(cp, n):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(3)]
    al = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)]
    be = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)]
    def fn():
        outs = []
        for r in range(R):
            outs += ops.snake_alias_group(xs, al, be, filt, [torch.empty_like(x) for x in xs])
        return outs
    return fn
def victim_conv(cp, n, k):
    x = torch.randn(1, n, cp, generator=gen).to("cuda")
    w = PW.pack_conv(torch.randn(cp, cp, k, generator=gen) / math.sqrt(cp * k), cp, cp).to("cuda")
    b = torch.randn(cp, generator=gen).to("cuda")
    def fn():
        return [ops.conv(x, w, b, ksize=k, pad=k // 2) for r in range(R)]
    return fn
VICTIMS = [("snake_conv_group c=20 (fp32 matrix-core form, U tile) n=48000", victim_amp(20, 20, 48000)),
           ("snake_conv_group c=10 (vector form, U tile) n=96000", victim_amp(10, 12, 96000)),
           ("snake_alias_group cp=40 n=24000", victim_alias(40, 24000)),
           ("snake_alias_group cp=20 n=48000", victim_alias(20, 48000)),
           ("conv k=7 cp=40 n=24000", victim_conv(40, 24000, 7))]
for name, fn in VICTIMS:
    Bg = graph(fn)
    torch.cuda.synchronize()
    with torch.cuda.stream(Bg[2]):
        Bg[0].replay()
    Bg[2].synchronize()
    ref = [o.clone() for o in Bg[1]]
    bad, worst = 0, 0.0
    for rep in range(10):
        for g, o, s in (A, Bg):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = max(float((o - r).abs().max()) for o, r in zip(Bg[1], ref))
        bad += e > 0; worst = max(worst, e)
    print(f"[probe4] victim {name}: {bad}/10 replays beside the culprit differ, worst {worst:.3e}", flush=True)
