"""Folder-of-wavs conversion, the reference's ``svc_inference_batch.py`` (:15-52), as the multi-GPU form of the hot path.

The reference runs Whisper over every file, then starts ``python svc_inference.py`` once per file (each child reloads
HuBERT, CREPE and the synthesizer).  Here the four models are loaded ONCE per process, and with ``torchrun --nproc-per-node N``
the files are assigned to the N ranks by longest-processing-time (cost = file size ~ duration): one process per GPU, rank 0
reads the checkpoints and ships them to the other ranks as one flat RCCL broadcast each (``svcmi.dist``), no collective
afterwards -- utterances are independent.  Outputs land in ``./_svc_out/<file>`` like the reference's.

    python -m svcmi.svc_inference_batch --config configs/base.yaml --model sovits5.0.pth --wave test_waves/ --spk singer.npy
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m svcmi.svc_inference_batch ...
"""
import os
import threading
import time

import numpy as np
import torch

from . import dist as D

OUT_PATH = "./_svc_out"


def list_waves(wave_path):
    """:29-31 (sorted, so every rank sees the same order)."""
    assert os.path.isdir(wave_path), f"{wave_path} is not folder"
    return sorted(f for f in os.listdir(wave_path) if f.endswith(".wav"))


def _load_on_rank0(path, rank):
    return torch.load(path, map_location="cpu") if rank == 0 else None


class Converter:
    """The four models of the wav -> wav path, resident on one GPU."""

    def __init__(self, args, device, rank=0, world=1):
        from .hubert import inference as hubert_inf
        from .pitch import inference as pitch_inf
        from .svc_inference import load_config
        from .vits.models import SynthesizerInfer
        from .whisper import inference as whisper_inf
        self.args, self.device = args, device
        self.hp = load_config(args.config)
        bc = lambda sd: D.broadcast_state_dict(sd, 0, device) if world > 1 else sd
        wck = _load_on_rank0(args.whisper, rank)
        dims = [wck["dims"] if rank == 0 else None]
        if world > 1:
            torch.distributed.broadcast_object_list(dims, src=0)
        self.whisper = whisper_inf.load_model({"dims": dims[0], "model_state_dict": bc(wck["model_state_dict"] if rank == 0 else None)}, device)
        self.hubert = hubert_inf.load_model(bc(_load_on_rank0(args.hubert, rank)), device)
        self.crepe = pitch_inf.load_crepe(bc(_load_on_rank0(args.crepe, rank)), device)
        self.crepe.precision = None if getattr(args, "f0_precision", "f32") == "f32" else getattr(args, "f0_precision", "f32")
        ck = _load_on_rank0(args.model, rank)
        self.model = SynthesizerInfer(self.hp.data.filter_length // 2 + 1, self.hp.data.segment_size // self.hp.data.hop_length, self.hp)
        want = self.model.state_dict()
        if rank == 0:                       # svc_inference.py:61-74: report what the checkpoint lacks instead of silently keeping the init values
            for k in want:
                if k not in ck["model_g"]:
                    print("%s is not in the checkpoint" % k)
        sd = bc({k: v for k, v in ck["model_g"].items() if k in want} if rank == 0 else None)
        self.model.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=False)
        self.model.eval()
        self.model.to(device)
        prec = None if getattr(args, "precision", "f32") == "f32" else args.precision
        self.whisper.encoder.precision = self.hubert.precision = "f16" if prec == "mixed" else prec     # (the mixed policy's classes are the synthesizer's)
        self.model.precision = prec
        self.spk = torch.FloatTensor(np.load(args.spk))
        self.tmp = os.path.join(OUT_PATH, f".rank{rank}")
        # everything lazily built (packed synthesizer weights, C model structs) is built HERE, on the constructing thread, and the device
        # is drained: the worker threads of run_batch start on their own streams with all shared state complete and visible
        self.model.warm()
        self.whisper.encoder._cmodel()
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    def convert(self, wav_path):
        """One file through PPG / vec / F0 extraction and svc_infer; returns np.float32 audio at hp.data.sampling_rate."""
        from .hubert import inference as hubert_inf
        from .pitch import inference as pitch_inf
        from .svc_inference import DummyRetrieval, shift_pitch, svc_infer
        from .whisper import inference as whisper_inf
        from .svc_inference import extract_features
        from .whisper.audio import load_audio
        # the three extractors in flight together, features kept on the device (the reference's per-file .npy / .csv intermediates
        # carry the same values: float32 arrays and the int()-quantised F0 of the pitch CSV, svc_inference.py:150-154,183)
        ppg, vec, pit = extract_features(load_audio(wav_path), self.whisper, self.hubert, self.crepe, self.device)
        ppg = torch.repeat_interleave(ppg, 2, 0)              # np.repeat(ppg, 2, 0), svc_inference.py:175-182
        vec = torch.repeat_interleave(vec, 2, 0)
        pit = torch.FloatTensor(shift_pitch(pitch_inf.quantize_pitch_like_csv(pit), self.args.shift))
        return svc_infer(self.model, DummyRetrieval(), self.spk, pit, ppg, vec, self.hp, self.device, write_pit_wav=False)


def run_batch(args, converter_factory=Converter, backend=None):
    """Shard the folder over the ranks, convert, write ``_svc_out/<file>``.  Returns this rank's file list."""
    from scipy.io.wavfile import write
    rank, local_rank, world = D.init_from_env(backend=backend)
    os.makedirs(OUT_PATH, exist_ok=True)
    waves = list_waves(args.wave)
    cost = [os.path.getsize(os.path.join(args.wave, f)) for f in waves]
    mine = D.shard_utterances(cost, world)[rank]
    device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    conv = converter_factory(args, device, rank, world)
    t0 = time.perf_counter()
    sr = conv.hp.data.sampling_rate
    seconds = [0.0] * len(mine)

    def work(slot):
        out = conv.convert(os.path.join(args.wave, waves[mine[slot]]))
        write(os.path.join(OUT_PATH, waves[mine[slot]]), sr, out)
        seconds[slot] = len(out) / sr

    workers = max(1, min(int(getattr(args, "workers", 1) or 1), len(mine)))
    if workers == 1 or not torch.cuda.is_available():
        for slot in range(len(mine)):
            work(slot)
    else:
        # files in flight (svcmi/lanes.py): every worker thread converts its files on its own HIP stream, so one file's launch chains
        # (and its host-side work: file I/O, CSV quantisation, launch overhead) overlap another file's GEMMs
        nxt, lock, errors = [0], threading.Lock(), []

        def loop():
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=device)):
                    while True:
                        with lock:
                            slot = nxt[0]
                            nxt[0] += 1
                        if slot >= len(mine) or errors:
                            return
                        work(slot)                       # (convert() ends with a D2H copy: the stream is drained)
            except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
                errors.append(e)

        threads = [threading.Thread(target=loop, name=f"svcmi-worker-{k}") for k in range(workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    audio_s = sum(seconds)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    rates = D.gather_stats(audio_s / max(time.perf_counter() - t0, 1e-9))
    if rank == 0:
        print(f"converted {len(waves)} files on {world} rank(s): {sum(rates):.1f} audio-seconds/s ({', '.join(f'{r:.1f}' for r in rates)})")
    return [waves[i] for i in mine]


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="svcmi drop-in for the reference's svc_inference_batch.py")
    p.add_argument("--config", type=str, required=True, help="yaml file for config.")
    p.add_argument("--model", type=str, required=True, help="path of model for evaluation")
    p.add_argument("--wave", type=str, required=True, help="Path of raw audio (a folder of .wav files).")
    p.add_argument("--spk", type=str, required=True, help="Path of speaker.")
    p.add_argument("--shift", type=int, default=0, help="Pitch shift key.")
    p.add_argument("--whisper", type=str, default=os.path.join("whisper_pretrain", "large-v2.pt"))
    p.add_argument("--hubert", type=str, default=os.path.join("hubert_pretrain", "hubert-soft-0d54a1f4.pt"))
    p.add_argument("--crepe", type=str, default=os.path.join("crepe", "assets", "full.pth"))
    p.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "f16", "mixed"],
                   help="GEMM operand precision of Whisper / HuBERT / synthesizer (see svc_inference); f32 = parity default")
    p.add_argument("--f0-precision", default="f32", choices=["f32", "bf16x3", "f16", "bf16"], help="GEMM operand precision of the CREPE F0 extractor (see svc_inference)")
    p.add_argument("--workers", type=int, default=3,
                   help="files in flight per GPU: worker threads, each converting its files on its own HIP stream (1 = the reference's order)")
    return p


if __name__ == "__main__":
    from svcmi.lanes import want_hw_queues
    want_hw_queues()                      # before the first HIP call: the chunk streams of svc_infer get their own hardware queues
    run_batch(build_parser().parse_args())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
