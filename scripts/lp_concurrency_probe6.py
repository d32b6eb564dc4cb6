"""Where in the victim's WORKSPACE do the first differences appear when a reduced-precision synthesizer graph runs beside it?"""
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
lens = [torch.tensor([T], dtype=torch.int32, device="cuda") for _ in range(2)]
def build(m, d, ln, stop=None):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    def fn():
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        m._stop_after = stop
        o = m.inference_ppg50(d["ppg"][:, ::2].contiguous(), d["vec"], d["pit"], d["spk"], ln, src, noise=d["enc_noise"])
        m._stop_after = None
        return o
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o, s
ms[0].precision = pol(amp3="f16")
A = build(ms[0], ds[0], lens[0])
Bg = build(ms[1], ds[1], lens[1], ("stage", 2))
key = [k for k in ops.workspaces if k[1] == Bg[2].cuda_stream and k[2] == "stage"][0]
ws = ops.workspaces[key]
base_off = ((ws.data_ptr() + 255) & ~255) - ws.data_ptr()
torch.cuda.synchronize()
ws.zero_()
with torch.cuda.stream(Bg[2]):
    Bg[0].replay()
Bg[2].synchronize()
ref = ws.clone()
n, cp = 24000, 40
sz = n * cp * 4
names = ["y", "acc"] + [f"xj{j}" for j in range(3)] + [f"t1_{j}" for j in range(3)] + [f"t2_{j}" for j in range(3)]
off0 = 73715200
for rep in range(4):
    ws.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(A[2]):
        A[0].replay()
    with torch.cuda.stream(Bg[2]):
        Bg[0].replay()
    torch.cuda.synchronize()
    for i, nm in enumerate(names):
        o = base_off + off0 + i * sz
        a = ws[o:o + sz].view(torch.float32).view(n, cp); r = ref[o:o + sz].view(torch.float32).view(n, cp)
        dm = a != r
        if bool(dm.any()):
            rows = torch.nonzero(dm.any(dim=1)).flatten()
            cols = torch.nonzero(dm.any(dim=0)).flatten()
            rl = rows.tolist()
            # runs of consecutive rows
            runs, st, pv = [], rl[0], rl[0]
            for x in rl[1:]:
                if x != pv + 1:
                    runs.append((st, pv)); st = x
                pv = x
            runs.append((st, pv))
            print(f"[probe6] rep {rep} {nm}: {int(dm.sum())} values differ, {len(rl)} rows in {len(runs)} runs {runs[:8]}, cols {cols.tolist()[:12]}{'...' if cols.numel() > 12 else ''}, max |diff| {float((a - r).abs().max()):.3e}, ref scale {float(r.abs().max()):.2f}; any NaN {bool(torch.isnan(a).any())}", flush=True)
