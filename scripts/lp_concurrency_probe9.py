"""Culprit = ONE reduced-precision kernel repeated for several ms; victim = the fp32 synthesizer graph (and a dependent fp32 chain)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops, weights as PW
from tests import engine_cases as E
from workload import config as C, inputs as I, weights as W
ops = Ops(); hp = C.base_hp(); T = 300
gen = torch.Generator().manual_seed(5)
filt = W.kaiser_sinc_filter().view(-1).to("cuda")
R = int(os.environ.get("PROBE_R", 80))
def graph(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
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
def culprit_amp(c, ld, n, precision):
    probs = amp_problems(c, ld, n)
    outs = [torch.empty_like(p["x"]) for p in probs]
    def fn():
        for r in range(R):
            ops.snake_conv_group([dict(p, out=o) for p, o in zip(probs, outs)], filt, c=c, precision=precision)
        return outs
    return fn
RV = int(os.environ.get("PROBE_RV", 40))
n2 = 24000
def v_conv_group(cp=40, n=n2, tile=0):
    probs = []
    for k in (3, 7, 11):
        x = torch.randn(1, n, cp, generator=gen).to("cuda")
        w = PW.pack_conv(torch.randn(cp, cp, k, generator=gen) / math.sqrt(cp * k), cp, cp).to("cuda")
        b = torch.randn(cp, generator=gen).to("cuda")
        res = torch.randn(1, n, cp, generator=gen).to("cuda")
        probs.append(dict(x=x, w=w, bias=b, ksize=k, pad=k // 2, res=res, tile=tile))
    outs = [[torch.empty(1, n, cp, device="cuda") for _ in range(3)] for _ in range(RV)]
    def fn():
        for r in range(RV):
            ops.conv_group([dict(p, out=o) for p, o in zip(probs, outs[r])])
        return [o for oo in outs for o in oo]
    return fn
def v_alias(cp=40, n=n2):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(3)]
    al = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)]
    be = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)]
    outs = [[torch.empty_like(x) for x in xs] for _ in range(RV)]
    def fn():
        for r in range(RV):
            ops.snake_alias_group(xs, al, be, filt, outs[r])
        return [o for oo in outs for o in oo]
    return fn
def v_conv(cp, n, k, cin=None, stride=1, nout=None, accumulate=False):
    cin = cin or cp; nout = nout or cp
    x = torch.randn(1, n * stride, cin, generator=gen).to("cuda")
    w = PW.pack_conv(torch.randn(nout, cin, k, generator=gen) / math.sqrt(cin * k), cin, nout).to("cuda")
    b = torch.randn(nout, generator=gen).to("cuda")
    y0 = torch.randn(1, n, nout, generator=gen).to("cuda")
    outs = [torch.empty(1, n, nout, device="cuda") for _ in range(RV)]
    def fn():
        for r in range(RV):
            if accumulate:
                outs[r].copy_(y0)
            ops.conv(x, w, b, ksize=k, stride=stride, pad=(k - stride) // 2 if stride > 1 else k // 2, t_out=n, accumulate=accumulate, out=outs[r])
        return outs
    return fn
def v_amp(c, ld, n):
    probs = amp_problems(c, ld, n)
    outs = [[torch.empty_like(p["x"]) for p in probs] for _ in range(RV)]
    def fn():
        for r in range(RV):
            ops.snake_conv_group([dict(p, out=o) for p, o in zip(probs, outs[r])], filt, c=c)
        return [o for oo in outs for o in oo]
    return fn
def v_mean(cp=40, n=n2):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(3)]
    outs = [torch.empty_like(xs[0]) for _ in range(RV)]
    def fn():
        for r in range(RV):
            ops.block_mean(xs, out=outs[r])
        return outs
    return fn
Ag = graph(culprit_amp(20, 20, 48000, "f16w2"))
VICTIMS = [("conv_group 40 ch (P16 64x48, 2-deep ring)", v_conv_group()), ("conv_group 80 ch n=6000", v_conv_group(80, 6000)), ("conv_group 160 ch n=1500", v_conv_group(160, 1500)),
           ("snake_alias_group 40 ch", v_alias()), ("conv k=7 40 ch", v_conv(40, n2, 7)), ("noise conv c_in=1 k=8 stride 4 accumulate", v_conv(40, n2, 8, cin=1, stride=4, accumulate=True)),
           ("up conv 80 -> 160 n=6000", v_conv(160, 6000, 3, cin=80, nout=160)), ("block_mean", v_mean()),
           ("snake_conv_group fp32 c=20 n=48000", v_amp(20, 20, 48000)), ("snake_conv_group fp32 c=10 n=96000", v_amp(10, 12, 96000))]
for name, fn in VICTIMS:
    Bg = graph(fn)
    torch.cuda.synchronize()
    with torch.cuda.stream(Bg[2]):
        Bg[0].replay()
    Bg[2].synchronize()
    ref = [o.clone() for o in Bg[1]]
    bad, worst = 0, 0.0
    for rep in range(6):
        for g, o, s in (Ag, Bg):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = max(float((o - r).abs().max()) for o, r in zip(Bg[1], ref))
        bad += e > 0; worst = max(worst, e)
    print(f"[probe9] victim {name} x{RV} beside snake_conv_group c=20 f16w2 x{R}: differs in {bad}/6 replays, worst {worst:.3e}", flush=True)
    del Bg, ref
