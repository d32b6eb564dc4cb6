"""Serving many fixed-shape conversion requests on one GPU: ``ClipLanes`` = N clips in flight for one shape bucket.

The reference converts one clip at a time (svc_inference.py:137-203).  A server that holds the models resident sees a queue of independent
requests; at batch 1 each of them is a chain of ~270 mostly latency-bound launches, and running N chains on N HIP streams (svcmi/lanes.py)
raises the throughput of the SAME batch-1 kernels by a third.  A bucket fixes the shape (B clips of T frames, 10 ms
each) so that every lane can be captured once as a HIP graph over static buffers; a request is staged into the next lane's buffers and
replayed.  ``bench.py`` (configs[1]) times exactly ``ClipLanes.launch``.
"""
import os

import torch

from .lanes import GraphLanes

INPUTS = ("mel", "vec", "pit", "spk")
# Which single-launch fp32 GEMMs of a lane's graph take the 2-deep operand ring (SVCMI_CONV_RING2: 41 instead of 61 KB of LDS per block
# = one more resident block per CU) when SEVERAL lanes share the chip -- bit mask: 1 Whisper QKV, 2 out-projection, 4 MLP-up, 8 MLP-down,
# 16 the synthesizer's GEMMs.  Measured on MI355X with 4 clips in flight (profiles/r05a_lanes_ring.log, r05b_ring2_sweep.log; audio-s/s
# of the judged line, box noise +-0.5 %): none 1395-1405, all 1426, Whisper only 1430, synthesizer only 1402, QKV + out-projection 1408,
# out-projection + MLP-down 1403, MLP-up + MLP-down 1438 -- the gain is the MLP-up launch (512 blocks of 61 KB = exactly two per CU) --
# while a clip that has the chip to itself loses 4 % with every class on: one-lane captures keep the 3-deep ring.  The ring depth does
# not change a single bit of the results.  SVCMI_RING2=<mask> overrides (tuning runs).
RING2_IN_FLIGHT = 12


def convert_step(model, whisper, buf, keep, noise=None):
    """One conversion on the tensors of ``buf``: mel [B,80,T] -(+0.1*randn)-> Whisper encoder -> first ``keep`` PPG frames ->
    pitch2source + prior encoder + reverse flow + NSF-BigVGAN -> [B,1,hop*T] (whisper/inference.py:32-62 + vits/models.py:251-256).
    ``noise`` = dict(mel_noise, rand_ini, src_noise, enc_noise) pins the stochastic draws; None draws on the device."""
    mel_noise = torch.randn_like(buf["mel"]) if noise is None else noise["mel_noise"]
    ppg50 = whisper.encoder(buf["mel"], mel_noise, 0.1)[:, :keep]
    src = model.pitch2source(buf["pit"], noise=None if noise is None else (noise["rand_ini"], noise["src_noise"]))
    return model.inference_ppg50(ppg50, buf["vec"], buf["pit"], buf["spk"], buf["lengths"], src,
                                 noise=None if noise is None else noise["enc_noise"])


class ClipLanes:
    """``lanes`` clips in flight for requests of B clips x T frames.  ``submit`` stages a request into the next lane and launches it,
    ``result`` waits for that lane and returns the waveform [B,1,hop*T]; at most ``lanes`` requests are outstanding (submitting into a lane
    whose result was not collected overwrites it in stream order)."""

    def __init__(self, model, whisper, T, B=1, lanes=4, device="cuda", pinned_noise=False, ring2=None):
        hp = model.hp
        env = os.environ.get("SVCMI_RING2")
        self.ring2 = int(ring2) if ring2 is not None else (int(env) if env is not None else (RING2_IN_FLIGHT if lanes > 1 else 0))
        self.model, self.whisper, self.T, self.B = model, whisper, int(T), int(B)
        self.keep = self.T // 2                                  # whisper/inference.py:40: len // 320 frames of 20 ms
        dev = torch.device(device)
        hop = hp.data.hop_length
        self.bufs, self.noises = [], []
        for _ in range(lanes):
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            self.bufs.append(dict(mel=z(B, 80, T), vec=z(B, T, hp.vits.vec_dim), pit=z(B, T), spk=z(B, hp.vits.spk_dim),
                                  lengths=torch.full((B,), T, dtype=torch.int32, device=dev)))
            self.noises.append(dict(mel_noise=z(B, 80, T), rand_ini=z(B, 11), src_noise=z(B, T * hop, 11),
                                    enc_noise=z(B, hp.vits.inter_channels, T)) if pinned_noise else None)
        self.lanes = None
        self._fns = [self._step_fn(i) for i in range(lanes)]

    def _step_fn(self, i):
        return lambda: convert_step(self.model, self.whisper, self.bufs[i], self.keep, self.noises[i])

    def stage(self, lane, noise=None, lengths=None, **inputs):
        """Copy a request into lane ``lane``'s static buffers (on that lane's stream once captured; before capture on the current one)."""
        stream = self.lanes.streams[lane] if self.lanes is not None else torch.cuda.current_stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for k in INPUTS:
                src = inputs[k].to(self.bufs[lane][k].device, torch.float32)
                self.bufs[lane][k].copy_(src, non_blocking=True)
                if src.is_cuda:
                    src.record_stream(stream)        # the caller may drop its tensor right after submit(): keep the block until the copy ran
            if lengths is not None:
                self.bufs[lane]["lengths"].copy_(torch.as_tensor(lengths, dtype=torch.int32), non_blocking=True)
            if noise is not None:
                if self.noises[lane] is None:
                    raise ValueError("construct ClipLanes(pinned_noise=True) to pass explicit noise")
                for k, v in noise.items():
                    self.noises[lane][k].copy_(v.to(self.noises[lane][k].device, torch.float32), non_blocking=True)

    def capture(self):
        """Warm up and capture every lane on its own stream (call after the first inputs are staged: the warm-up runs on them)."""
        if self.lanes is None:
            import ctypes
            ops = self.model.ops
            lib = ops.lib
            with ops._lock:                                # the knob is process-wide: no other thread of this Ops captures or launches meanwhile
                prev = ctypes.c_int32(0)
                if lib.svcmi_tune_get(b"ring2", ctypes.byref(prev)) != 0:
                    raise ValueError("svcmi_tune_get(ring2)")
                if self.ring2 != prev.value and lib.svcmi_tune_set(b"ring2", self.ring2) != 0:
                    raise ValueError(f"ring2 mask {self.ring2}")
                try:
                    self.lanes = GraphLanes(self._fns)         # the kernels chosen during capture are what every replay runs
                finally:
                    lib.svcmi_tune_set(b"ring2", prev.value)   # what the host had set (svcmi_tune_set / SVCMI_TUNE), not a default
        return self

    def launch(self, lane=None):
        """Replay the next lane (round-robin) on whatever its buffers hold; returns the lane index."""
        return self.capture().lanes.launch(lane)

    def submit(self, noise=None, lengths=None, **inputs):
        self.capture()
        lane = self.lanes._next
        self.stage(lane, noise=noise, lengths=lengths, **inputs)
        return self.lanes.launch(lane)

    def result(self, lane, clone=True):
        out = self.lanes.wait(lane)
        return out.clone() if clone else out

    def synchronize(self):
        if self.lanes is not None:
            self.lanes.synchronize()
