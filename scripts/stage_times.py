#!/usr/bin/env python
"""In-situ wall time of each pipeline segment on the GPU box: the bench step is captured as HIP graphs that stop after
successive segments (whisper | prior encoder | flow | generator pre | stage 0..4 | output layer) and the replay times are
differenced.  Unlike a rocprofv3 trace this measures the un-instrumented graph, multi-stream overlap included.
Usage: python scripts/stage_times.py [--ungrouped [--streams]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
from svcmi import Ops  # noqa: E402
from workload import config as C  # noqa: E402
from workload import weights as W  # noqa: E402


def replay_ms(g, iters=30):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ops = Ops()
    hp = C.base_hp()
    wl = bench.Workload(ops, "cuda", 1, 10.0, W.make_whisper_state(C.WHISPER_LARGE_V2), W.make_vits_state(hp, seed=1234), hp, seed=100)
    if "--ungrouped" in sys.argv:        # the pre-r01i structure: one launch per AMP block and step (serial, or forked streams)
        wl.model.grouped_blocks = False
        wl.model.parallel_blocks = "--streams" in sys.argv
    full_step = wl.step

    def whisper_only():
        mel_noise = torch.randn_like(wl.mel)
        return wl.whisper.encoder(wl.mel, mel_noise, 0.1)[:, :wl.keep]

    stops = [("whisper", None), ("prior", "prior"), ("flow", "flow"), ("gen_pre", "gen_pre")] + \
            [(f"stage{i}", ("stage", i)) for i in range(5)] + [("post", "full")]
    prev, rows = 0.0, []
    for name, stop in stops:
        if name == "whisper":
            wl.step = whisper_only
        else:
            wl.step = full_step
            wl.model._stop_after = None if stop == "full" else stop
        g, _ = bench.build_graph(wl)
        ms = replay_ms(g)
        rows.append((name, ms, ms - prev))
        prev = ms
        del g
    for name, ms, d in rows:
        print(f"{name:8s} cumulative {ms:7.3f} ms   segment {d:6.3f} ms", flush=True)


if __name__ == "__main__":
    main()
