"""Victims = single fp32 kernels of generator stage 2 (40 channels) repeated in a HIP graph, beside the synthesizer with its 20-channel stage on
the fp16 matrix cores (launched first): which one changes?"""
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
m = E.make_model(hp, ops, "cuda")[0]
m.precision = pol(amp3="f16") if len(sys.argv) < 2 else (None if sys.argv[1] == "f32" else pol(**dict(a.split("=") for a in sys.argv[1:])))
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
R = int(os.environ.get("PROBE_R", 60))
n = 24000
def v_conv_group(cp=40):
    probs = []
    for k in (3, 7, 11):
        x = torch.randn(1, n, cp, generator=gen).to("cuda")
        w = PW.pack_conv(torch.randn(cp, cp, k, generator=gen) / math.sqrt(cp * k), cp, cp).to("cuda")
        b = torch.randn(cp, generator=gen).to("cuda")
        res = torch.randn(1, n, cp, generator=gen).to("cuda")
        probs.append(dict(x=x, w=w, bias=b, ksize=k, pad=k // 2, res=res))
    def fn():
        outs = []
        for r in range(R):
            outs += ops.conv_group([dict(p, out=torch.empty(1, n, cp, device="cuda")) for p in probs])
        return outs
    return fn
def v_noise_conv(cp=40, k=8, stride=4, split_k=0):
    src = torch.randn(1, n * stride, 1, generator=gen).to("cuda")
    w = PW.pack_conv(torch.randn(cp, 1, k, generator=gen) / math.sqrt(k), 1, cp).to("cuda")
    b = torch.randn(cp, generator=gen).to("cuda")
    y0 = torch.randn(1, n, cp, generator=gen).to("cuda")
    def fn():
        outs = []
        for r in range(R):
            y = y0.clone()
            outs.append(ops.conv(src, w, b, ksize=k, stride=stride, pad=(k - stride) // 2, t_out=n, accumulate=True, out=y, split_k=split_k))
        return outs
    return fn
def v_up_conv(cin=80, cp=40, u=4, taps=2, split_k=0):
    x = torch.randn(1, n // u, cin, generator=gen).to("cuda")
    w = PW.pack_conv(torch.randn(u * cp, cin, taps, generator=gen) / math.sqrt(cin * taps), cin, u * cp).to("cuda")
    b = torch.randn(u * cp, generator=gen).to("cuda")
    def fn():
        return [ops.conv(x, w, b, ksize=taps, pad=taps // 2, t_out=n // u, split_k=split_k) for r in range(R)]
    return fn
def v_block_mean(cp=40):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(3)]
    return lambda: [ops.block_mean(xs) for r in range(R)]
VICTIMS = [("conv_group 3/7/11 taps, 40 channels, n = 24000", v_conv_group()),
           ("noise conv c_in = 1 (scalar gather), k 8 stride 4, accumulate, split-K heuristic", v_noise_conv()),
           ("noise conv ..., split_k = 1", v_noise_conv(split_k=1)),
           ("up conv 80 -> 4 x 40 polyphase, split-K heuristic", v_up_conv()),
           ("up conv ..., split_k = 1", v_up_conv(split_k=1)),
           ("block_mean", v_block_mean())]
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
    print(f"[probe7] victim {name}: {bad}/10 replays beside the culprit differ, worst {worst:.3e}", flush=True)
