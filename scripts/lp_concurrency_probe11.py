"""Lane pattern of the wrong SnakeAlias values (victim beside the fp16 fused half-step): for each wrong output tensor the (wave, lanes, out index) sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ["PROBE_SET"] = "none"
import importlib.util, torch, collections
spec = importlib.util.spec_from_file_location("p10", os.path.join(ROOT, "scripts", "lp_concurrency_probe10.py"))
# (re-uses the builders of probe10 without running its trials)
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
events = 0
for rep in range(6):
    for G, o, s in (Ag, Bg):
        with torch.cuda.stream(s): G.replay()
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(Bg[1], ref)):
        dm = (o[0] != r[0])
        if not bool(dm.any()) or events >= 14: continue
        events += 1
        idx = torch.nonzero(dm).tolist()
        by_wave = collections.defaultdict(lambda: collections.defaultdict(set))
        for t, ch in idx:
            e = (t // 8) * cp + ch
            by_wave[e // 64][t % 8].add(e % 64)
        desc = []
        for wv in sorted(by_wave):
            for oi in sorted(by_wave[wv]):
                ln = sorted(by_wave[wv][oi])
                desc.append(f"wave {wv} (block {wv // 4}, wave-in-block {wv % 4}) out[{oi}] lanes {ln[0]}-{ln[-1]} ({len(ln)})" if ln == list(range(ln[0], ln[-1] + 1)) else f"wave {wv} out[{oi}] lanes {ln}")
        print(f"rep {rep} launch {i // 3} problem {i % 3}: {len(idx)} wrong values in {len(by_wave)} waves")
        for dsc in desc[:24]: print("    ", dsc)
        if len(desc) > 24: print("     ...", len(desc), "entries")
