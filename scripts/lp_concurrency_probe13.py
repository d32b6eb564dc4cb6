"""Saves one wrong SnakeAlias output (victim beside the fp16 fused half-step) with its input and parameters for offline algebra: gpurun_out/r06y/alias_event.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = open(os.path.join(ROOT, "scripts", "lp_concurrency_probe10.py")).read().split("which = os.environ.get")[0]
g = {"__name__": "p10", "__file__": os.path.join(ROOT, "scripts", "lp_concurrency_probe10.py")}
exec(compile(src, "p10", "exec"), g)
cf, _ = g["culprit_amp"](20, 20, 48000, "f16w2")
cp, n = 40, 24000
va, vin = g["v_alias"](cp, n)
Ag = g["graph"](cf); Bg = g["graph"](va)
torch.cuda.synchronize()
with torch.cuda.stream(Bg[2]): Bg[0].replay()
Bg[2].synchronize()
ref = [o.clone() for o in Bg[1]]
saved = 0
for rep in range(6):
    for G, o, s in (Ag, Bg):
        with torch.cuda.stream(s): G.replay()
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(Bg[1], ref)):
        if saved or torch.equal(o, r): continue
        j = i % 3
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r06y", "alias_event.npz"), x=vin[j].cpu().numpy(), al=vin[3 + j].cpu().numpy(), be=vin[6 + j].cpu().numpy(),
                            filt=g["filt"].cpu().numpy(), got=o.cpu().numpy(), ref=r.cpu().numpy())
        saved = 1
        print("saved event: launch", i // 3, "problem", j, "wrong values", int((o != r).sum()))
