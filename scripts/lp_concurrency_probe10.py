"""What exactly goes wrong in a SnakeAlias launch that runs beside the fp16 fused half-step?  Pattern of the wrong values, input integrity,
torch's own kernels as victims, culprit / victim variants through the tuning knobs."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
import torch
from svcmi import Ops, weights as PW
from workload import weights as W
ops = Ops()
def tune(k, v):
    assert ops.lib.svcmi_tune_set(k.encode(), int(v)) == 0, k
gen = torch.Generator().manual_seed(5)
filt = W.kaiser_sinc_filter().view(-1).to("cuda")
R = int(os.environ.get("PROBE_R", 80)); RV = int(os.environ.get("PROBE_RV", 40))
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
    return fn, probs
def v_alias(cp=40, n=24000, count=3):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(count)]
    al = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(count)]
    be = [(torch.randn(cp, generator=gen) * 0.3).to("cuda") for _ in range(count)]
    outs = [[torch.empty_like(x) for x in xs] for _ in range(RV)]
    def fn():
        for r in range(RV):
            ops.snake_alias_group(xs, al, be, filt, outs[r])
        return [o for oo in outs for o in oo]
    return fn, xs + al + be
def v_torch(kind, cp=40, n=24000):
    xs = [torch.randn(1, n, cp, generator=gen).to("cuda") for _ in range(3)]
    outs = [[torch.empty_like(x) for x in xs] for _ in range(RV)]
    def fn():
        for r in range(RV):
            for x, o in zip(xs, outs[r]):
                if kind == "copy": o.copy_(x)
                elif kind == "sin": torch.sin(x, out=o)
                else:
                    torch.mul(x, 1.25, out=o); o.add_(x).sin_().mul_(x)
        return [o for oo in outs for o in oo]
    return fn, xs
def runs_of(rl):
    runs, st, pv = [], rl[0], rl[0]
    for x in rl[1:]:
        if x != pv + 1:
            runs.append((st, pv)); st = x
        pv = x
    runs.append((st, pv))
    return runs
def trial(tag, culprit, victim, inputs, reps=int(os.environ.get("PROBE_REPS", 4)), dump=0):
    Ag = graph(culprit); Bg = graph(victim)
    torch.cuda.synchronize()
    with torch.cuda.stream(Bg[2]):
        Bg[0].replay()
    Bg[2].synchronize()
    ref = [o.clone() for o in Bg[1]]
    in0 = [t.clone() for t in inputs]
    with torch.cuda.stream(Ag[2]):
        Ag[0].replay()
    Ag[2].synchronize()
    cref = [o.clone() for o in Ag[1]]
    bad, worst, cbad = 0, 0.0, 0
    for rep in range(reps):
        for g, o, s in (Ag, Bg):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        errs = [float((o - r).abs().max()) for o, r in zip(Bg[1], ref)]
        cbad += any(not torch.equal(o, r) for o, r in zip(Ag[1], cref))
        e = max(errs); bad += e > 0; worst = max(worst, e)
        if dump and e > 0 and rep < 2:
            shown = 0
            for i, (o, r) in enumerate(zip(Bg[1], ref)):
                if errs[i] == 0 or shown >= dump: continue
                shown += 1
                a, b = o[0], r[0]
                dm = a != b
                rows = torch.nonzero(dm.any(dim=1)).flatten().tolist(); cols = torch.nonzero(dm.any(dim=0)).flatten().tolist()
                rr = runs_of(rows)
                print(f"   rep {rep} out[{i // 3}][{i % 3}]: {int(dm.sum())} values, rows {rr[:10]}{'...' if len(rr) > 10 else ''} ({len(rows)} rows), {len(cols)} cols {cols[:6]}..; |got| max {float(a[dm].abs().max()):.3e} ref {float(b[dm].abs().max()):.3e}; NaN {bool(torch.isnan(a).any())}; zeros among wrong {int((a[dm] == 0).sum())}")
                t0, c0 = rows[0], cols[0]
                same_other = [j for j in range(len(ref)) if j % 3 == i % 3 and j != i and torch.equal(Bg[1][j], ref[j])]
                # do the wrong values of row t0 equal the reference at a shifted row (misplaced data) ?
                sh = [d for d in range(-16, 17) if 0 <= t0 + d < b.shape[0] and d != 0 and torch.equal(a[t0], b[t0 + d])]
                print(f"      row {t0}: got {a[t0, :4].tolist()} ref {b[t0, :4].tolist()}  equal to the reference of a shifted row: {sh}")
    intact = all(torch.equal(t, t0) for t, t0 in zip(inputs, in0))
    print(f"[probe10] {tag}: victim differs in {bad}/{reps} replays, worst {worst:.3e}; inputs intact {intact}; the CULPRIT's own outputs differ from its solo run in {cbad}/{reps}", flush=True)
    del Ag, Bg, ref
which = os.environ.get("PROBE_SET", "all")
cf, cprob = culprit_amp(20, 20, 48000, "f16w2")
va, vin = v_alias()
trial("alias 40 ch beside f16w2 c=20", cf, va, vin, dump=int(os.environ.get("PROBE_DUMP", 3)))
if which == "first": sys.exit(0)
for kind in ("copy", "sin", "chain"):
    vt, tin = v_torch(kind)
    trial(f"torch {kind} beside f16w2 c=20", cf, vt, tin)
for cp, n in ((20, 48000), (12, 96000), (80, 12000)):
    v2, vin2 = v_alias(cp, n)
    trial(f"alias {cp} ch n={n} beside f16w2 c=20", cf, v2, vin2)
v1, vin1 = v_alias(40, 24000, 1)
trial("alias 40 ch ONE tensor per launch beside f16w2 c=20", cf, v1, vin1)
for rt in (12, 16):
    tune("snake_rt", rt)
    trial(f"alias 40 ch snake_rt={rt} beside f16w2 c=20", cf, va, vin)
tune("snake_rt", 0)
tune("amp_u", -1)
trial("alias 40 ch beside f16w2 c=20 WITHOUT the U tile (amp_u=-1)", cf, va, vin)
tune("amp_u", 0)
cf1, _ = culprit_amp(20, 20, 48000, "f16")
trial("alias 40 ch beside f16 (one term) c=20", cf1, va, vin)
cf32, _ = culprit_amp(20, 20, 48000, None)
trial("alias 40 ch beside the fp32 matrix-core half-step c=20 (control)", cf32, va, vin)
tune("amp_mfma", 0)
trial("alias 40 ch beside the fp32 VECTOR half-step c=20 (control)", cf32, va, vin)
tune("amp_mfma", 1)
