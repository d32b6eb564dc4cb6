import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from svcmi import Ops
from svcmi.pitch import inference as PI
from workload import weights as W
ops = Ops()
crepe = PI.load_crepe(W.make_crepe_state("full"), "cuda", ops=ops)
n = 160000
wav = (0.4 * torch.sin(2 * np.pi * 220.0 * torch.arange(n) / 16000.0)).float()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): o = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, o
ms, prob = t(lambda: crepe.probabilities(wav, hop=320, batch_size=512)); print(f"probabilities (H2D + frames + 6 convs + classifier): {ms:.2f} ms", prob.shape)
lt = torch.from_numpy(np.log(PI._transition() + np.finfo(np.float32).tiny)).cuda()
ms, bins0 = t(lambda: ops.viterbi_decode(prob, lt, 512, PI._frequency_to_bins(50.0), PI._frequency_to_bins(1000.0, ceil=True))); print(f"viterbi_decode dense: {ms:.2f} ms")
ms, bins = t(lambda: ops.viterbi_decode(prob, lt, 512, PI._frequency_to_bins(50.0), PI._frequency_to_bins(1000.0, ceil=True), band=11)); print(f"viterbi_decode banded: {ms:.2f} ms  equal: {bool((bins == bins0).all())}")
ms, _ = t(lambda: PI.bins_to_hz(bins, None)); print(f"bins_to_hz: {ms:.2f} ms")
ms, _ = t(lambda: torch.randn_like(wav)); print(f"cpu randn: {ms:.2f} ms")
ms, _ = t(lambda: PI.compute_f0_sing(wav, "cuda", model=crepe)); print(f"compute_f0_sing total: {ms:.2f} ms")
