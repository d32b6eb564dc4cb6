#!/usr/bin/env python
"""The CREPE F0 extractor alone on a 10 s clip (for rocprofv3): python scripts/crepe_profile.py [precision] [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from svcmi import Ops  # noqa: E402
from svcmi.pitch import inference as PI  # noqa: E402
from workload import weights as W  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ops = Ops()
m = PI.Crepe(W.make_crepe_state("full"), "cuda", ops=ops)
m.precision = None if prec == "f32" else prec
g = torch.Generator().manual_seed(0)
audio = (0.3 * torch.sin(torch.arange(160000) * 0.05) + 0.02 * torch.randn(160000, generator=g)).float()
PI.compute_f0_sing(audio, "cuda", model=m)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    PI.compute_f0_sing(audio, "cuda", model=m)
torch.cuda.synchronize()
print(f"crepe full, {prec}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per 10 s clip")
