#!/usr/bin/env python
"""The folder driver (svcmi.svc_inference_batch = the reference's svc_inference_batch.py) at full model sizes on synthetic files:
N x 10 s wav files -> _svc_out/, audio-seconds per second, wav -> wav including file I/O.  python scripts/folder_probe.py [files] [workers]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from scipy.io import wavfile  # noqa: E402

from oracle import audio_oracle as A  # noqa: E402  (synthetic audio generator only)
from svcmi import svc_inference_batch as SB  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hp = C.base_hp()
    d = tempfile.mkdtemp(prefix="svcmi_folder_")
    os.chdir(d)
    os.makedirs("waves")
    for i in range(n_files):
        audio = (A.synth_audio(16000 * 10, 8 + i) * 0.5).numpy()
        wavfile.write(f"waves/u{i:03d}.wav", 16000, (audio * 32767).astype(np.int16))
    t0 = time.perf_counter()
    torch.save({"model_g": W.make_vits_state(hp, seed=1234)}, "svc.pth")
    torch.save(W.make_whisper_state(C.WHISPER_LARGE_V2), "whisper.pt")
    torch.save(W.make_hubert_state(), "hubert.pt")
    torch.save(W.make_crepe_state("full"), "crepe.pth")
    np.save("spk.npy", I.synth_spk(hp.vits.spk_dim, seed=7).numpy())
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(json.loads(json.dumps(hp)), f)
    print(f"checkpoints written in {time.perf_counter() - t0:.1f} s", flush=True)
    base = ["--config", "cfg.yaml", "--model", "svc.pth", "--wave", "waves", "--spk", "spk.npy", "--whisper", "whisper.pt",
            "--hubert", "hubert.pt", "--crepe", "crepe.pth"]
    for w in ([workers] if len(sys.argv) > 2 else [1, 1, 2, 3, 4, 6]):      # (the first pass also warms the file-system cache)
        args = SB.build_parser().parse_args(base + ["--workers", str(w)])
        t0 = time.perf_counter()
        SB.run_batch(args)
        print(f"workers {w}: run_batch wall {time.perf_counter() - t0:.1f} s for {n_files} files (incl. model load)", flush=True)


if __name__ == "__main__":
    main()
