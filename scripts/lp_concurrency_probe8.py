"""Victim = a CHAIN of dependent fp32 kernels that reuse their buffers (SnakeAlias -> grouped GEMM -> SnakeAlias -> grouped GEMM ..., the AMP
half-steps of generator stage 2 as the stage host issues them), beside the synthesizer with its 20-channel stage on the fp16 matrix cores."""
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
m.precision = None if (len(sys.argv) > 1 and sys.argv[1] == "f32") else pol(amp3="f16")
d = {k: v.to("cuda") for k, v in I.synth_clip(T=T, hp=hp, seed=80, B=1).items()}
lens = torch.tensor([T], dtype=torch.int32, device="cuda")
def graph(fn, warm=True):
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
cp, n, R = 40, 24000, int(os.environ.get("PROBE_R", 8))
x0 = [torch.randn(1, n, cp, generator=gen).to("cuda") * 0.5 for _ in range(3)]
al = [[(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)] for _ in range(2)]
be = [[(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(3)] for _ in range(2)]
ws = [[PW.pack_conv(torch.randn(cp, cp, k, generator=gen) / math.sqrt(cp * k) * 0.7, cp, cp).to("cuda") for k in (3, 7, 11)] for _ in range(2)]
bs = [[(torch.randn(cp, generator=gen) * 0.1).to("cuda") for _ in range(3)] for _ in range(2)]
# static buffers, reused by every half-step (as the stage host's arena does)
xj = [torch.empty(1, n, cp, device="cuda") for _ in range(3)]
t1 = [torch.empty(1, n, cp, device="cuda") for _ in range(3)]
t2 = [torch.empty(1, n, cp, device="cuda") for _ in range(3)]
acc = torch.empty(1, n, cp, device="cuda")
def chain():
    xc = x0
    for r in range(R):
        outs = xj if r % 2 == 0 else t2
        ops.snake_alias_group(xc, al[0], be[0], filt, t1)
        tgt = t2 if r % 2 == 0 else xj
        ops.conv_group([dict(x=t1[j], w=ws[0][j], bias=bs[0][j], ksize=(3, 7, 11)[j], pad=(3, 7, 11)[j] // 2, out=tgt[j], tile=4) for j in range(3)])
        ops.snake_alias_group(tgt, al[1], be[1], filt, t1)
        ops.conv_group([dict(x=t1[j], w=ws[1][j], bias=bs[1][j], ksize=(3, 7, 11)[j], pad=(3, 7, 11)[j] // 2, res=xc[j], out=outs[j], tile=4) for j in range(3)])
        xc = outs
    ops.block_mean(xc, out=acc)
    return [acc] + list(xc) + t1
Bg = graph(chain)
torch.cuda.synchronize()
with torch.cuda.stream(Bg[2]):
    Bg[0].replay()
Bg[2].synchronize()
ref = [o.clone() for o in Bg[1]]
print(f"[probe8] chain output scale {float(ref[0].abs().max()):.3f}", flush=True)
for first in ("culprit", "victim"):
    bad, worst = 0, 0.0
    for rep in range(10):
        order = (A, Bg) if first == "culprit" else (Bg, A)
        for g, o, s in order:
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        e = max(float((o - r).abs().max()) for o, r in zip(Bg[1], ref))
        bad += e > 0; worst = max(worst, e)
    print(f"[probe8] culprit = synthesizer ({'fp32' if m.precision is None else 'amp3=f16'}), {first} launched first: chain differs in {bad}/10 replays, worst {worst:.3e}", flush=True)
