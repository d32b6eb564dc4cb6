import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
import torch
from svcmi import Ops, SynthesizerInfer
from workload import config as C, inputs as I, weights as W
ops, dev, hp = Ops(), "cuda", C.base_hp()
m = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
m.load_state_dict(W.make_vits_state(hp, seed=1234)); m.eval(); m.to(dev)
d = I.synth_clip(T=1000, hp=hp, seed=0, B=1)
ppg, vec, pit, spk = d["ppg"].to(dev), d["vec"].to(dev), d["pit"].to(dev), d["spk"].to(dev)
def synth():
    src = m.pitch2source(pit)
    return m.inference(ppg, vec, pit, spk, torch.tensor([1000]), src)
for _ in range(3): synth()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); synth(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms, launches {ops.launches}")
pr = cProfile.Profile(); pr.enable(); synth(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
