"""Clips in flight: N independent capture lanes over ONE set of weights.

At batch 1 a clip is ~270 dependent launches, a third of them latency-bound (prior encoder, flow, the narrow generator stages: 8-17 us
each for a few hundred MFLOP) and every GEMM launch spends 8-10 us of its 20-65 us outside the matrix pipe (operand round trip, epilogue,
drain).  HIP has no programmatic dependent launch to overlap consecutive kernels of ONE chain, but kernels of DIFFERENT clips are
independent: with each clip's chain on its own HIP stream -- mapped to its own hardware queue -- the command processor fills one clip's
launch gaps, ramp-ups and tails with another clip's kernels (measured: per-launch fixed costs and launch chains overlap; two chip-filling
kernels of different clips still take turns -- DESIGN.md section 5).  A lane = a HIP stream + a HIP graph captured on it + the static input /
output tensors its step function closes over; the split-K workspace of ``Ops`` is keyed by stream, so lanes share nothing but the
read-only weights.  Measured on MI355X (profiles/r02o_inflight_sweep.log, r02p_inflight_sweep.log): 1026 -> 1379 audio-s/s at batch 1, fp32, with 4 lanes on 8 hardware
queues; per-clip results are bit-identical to a single-stream run (tests/test_gpu_engine.py).

The ROCm runtime maps streams onto ``GPU_MAX_HW_QUEUES`` hardware queues (default 4) in creation order, shared with every other stream the
process made; two lanes that land on one queue serialise.  ``want_hw_queues()`` raises the count (must run before the first HIP call).
"""
import os

import torch

DEFAULT_HW_QUEUES = 8


def want_hw_queues(n=DEFAULT_HW_QUEUES):
    """Ask the ROCm runtime for ``n`` hardware queues per process unless the user chose otherwise.  No effect once HIP is initialised."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(n))


class GraphLanes:
    """``step_fns[i]()`` runs one clip on lane i's static buffers and returns its output tensor(s).  Each lane is warmed up and
    captured on its own stream; ``launch()`` replays the next lane round-robin and returns its index; a lane's output is valid after
    ``wait(i)`` / ``synchronize()`` and until that lane is launched again."""

    def __init__(self, step_fns, warm=2):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphLanes needs a GPU (svcmi has no CPU path)")
        self.streams, self.graphs, self.outputs, self.done = [], [], [], []
        for fn in step_fns:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warm):
                    fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = fn()
            self.streams.append(s)
            self.graphs.append(g)
            self.outputs.append(out)
            self.done.append(torch.cuda.Event())
        torch.cuda.synchronize()
        self._next = 0

    def __len__(self):
        return len(self.graphs)

    def launch(self, lane=None):
        i = self._next if lane is None else lane
        self._next = (i + 1) % len(self.graphs)
        with torch.cuda.stream(self.streams[i]):
            self.graphs[i].replay()
            self.done[i].record()
        return i

    def wait(self, lane):
        self.done[lane].synchronize()
        return self.outputs[lane]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
