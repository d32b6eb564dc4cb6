"""Probe (round 6): batch-1 requests served as ONE batched Whisper launch set per 4 requests (rows of the 4 windows flattened: M = 2000, chip-filling
128-row tiles) + 4 synthesizer lanes, double-buffered PPG, against 4 whole-conversion lanes (ClipLanes, the judged regime)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
from svcmi.lanes import want_hw_queues
want_hw_queues(8)
import torch
from svcmi import Ops
from svcmi.serving import ClipLanes
from svcmi.whisper.inference import load_model
from tests import engine_cases as E
from workload import config as C, inputs as I, weights as W
ops = Ops(); hp = C.base_hp(); T = 1000; NB = int(os.environ.get("HYBRID_BATCH", 4)); dev = "cuda"
m, _ = E.make_model(hp, ops, dev)
wm = load_model(W.make_whisper_state(C.WHISPER_LARGE_V2), dev, ops=ops)
keep = T // 2
clips = [{k: v.to(dev) for k, v in I.synth_clip(T=T, hp=hp, seed=70 + i, B=1, ppg=False).items()} for i in range(NB)]
lens = torch.full((1,), T, dtype=torch.int32, device=dev)
K = int(os.environ.get("HYBRID_STEPS", 80))

def timed(run, sync, steps, warm=8):
    for _ in range(warm): run()
    sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): run()
    sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

# ---- reference regime: 4 whole-conversion lanes
cl = ClipLanes(m, wm, T, B=1, lanes=4, device=dev)
for i in range(4):
    c = clips[i % NB]; cl.stage(i, mel=c["mel"], vec=c["vec"], pit=c["pit"], spk=c["spk"], lengths=lens)
cl.capture()
dt = timed(cl.launch, cl.synchronize, K)
print(f"[hybrid] ClipLanes, 4 clips in flight: {dt * 1e3:.3f} ms per clip = {10.0 / dt:.1f} audio-s/s", flush=True)

# ---- hybrid: batched Whisper + synthesizer lanes
mel4 = torch.cat([c["mel"] for c in clips], 0).contiguous()                     # [NB, 80, T]
P = [torch.zeros(NB, keep, 1280, device=dev) for _ in range(2)]
sw = torch.cuda.Stream(); ss = [torch.cuda.Stream() for _ in range(NB)]
def wfn(b):
    def fn():
        ppg = wm.encoder(mel4, torch.randn_like(mel4), 0.1)[:, :keep]
        P[b].copy_(ppg)
        return P[b]
    return fn
def sfn(b, i):
    c = clips[i]
    def fn():
        src = m.pitch2source(c["pit"])
        return m.inference_ppg50(P[b][i:i + 1], c["vec"], c["pit"], c["spk"], lens, src)
    return fn
def capture(fn, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = fn()
    return g, o
Wg = [capture(wfn(b), sw) for b in range(2)]
Sg = [[capture(sfn(b, i), ss[i]) for i in range(NB)] for b in range(2)]
torch.cuda.synchronize()
ready = [torch.cuda.Event() for _ in range(2)]; done = [[torch.cuda.Event() for _ in range(NB)] for _ in range(2)]
state = {"k": 0}
def run_batch():
    b = state["k"] & 1; state["k"] += 1
    with torch.cuda.stream(sw):
        for i in range(NB): sw.wait_event(done[b][i])
        Wg[b][0].replay(); ready[b].record()
    for i in range(NB):
        with torch.cuda.stream(ss[i]):
            ss[i].wait_event(ready[b]); Sg[b][i][0].replay(); done[b][i].record()
def sync():
    sw.synchronize()
    for s in ss: s.synchronize()
for b in range(2):
    for i in range(NB): done[b][i].record()
dtb = timed(run_batch, sync, K // NB, warm=4)
print(f"[hybrid] batched Whisper ({NB} windows, M = {NB * 500}) + {NB} synthesizer lanes: {dtb / NB * 1e3:.3f} ms per clip = {10.0 * NB / dtb:.1f} audio-s/s", flush=True)
# the two stages alone
dW = timed(lambda: Wg[0][0].replay(), sw.synchronize, 20, warm=3)
def s_all():
    for i in range(NB):
        with torch.cuda.stream(ss[i]): Sg[0][i][0].replay()
dS = timed(s_all, sync, 20, warm=3)
print(f"[hybrid] alone: Whisper batch {dW * 1e3:.3f} ms ({dW / NB * 1e3:.3f} per clip), {NB} synthesizer lanes {dS * 1e3:.3f} ms ({dS / NB * 1e3:.3f} per clip)", flush=True)
# parity of the hybrid's waveforms against the ClipLanes ones is not checked here (different noise draws); per-item equality of the batched encoder: tests
