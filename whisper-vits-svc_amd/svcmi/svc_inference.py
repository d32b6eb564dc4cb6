"""Drop-in for the synthesis driver of the reference's svc_inference.py (:61-134): same function names,
argument order and results, with the chunk loop kept on the device.

    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp)
    load_svc_model(path, model); model.eval(); model.to("cuda")
    out = svc_infer(model, retrieval, spk, pit, ppg, vec, hp, "cuda")       # np.float32 [320*T - 1]

Reference behaviours that are preserved on purpose: inputs truncated to the shortest of pit/vec/ppg (:78-85);
2500-frame chunks with a 10-frame halo that is computed and discarded (:94-131) -- attention is global within
a chunk, so the schedule is part of the numerics; the final sample is dropped because the last chunk uses
``cut_e_out = -1`` (:112,129); ``svc_out_pit.wav`` side output (:91-92).
"""
import os

import numpy as np
import contextlib
import threading

import torch

from .vits import consts as K


class IRetrieval:
    """feature_retrieval/retrieval.py:11-28 hook: per-chunk transform of the content features."""

    def retriv_whisper(self, vec):
        raise NotImplementedError

    def retriv_hubert(self, vec):
        raise NotImplementedError


class DummyRetrieval(IRetrieval):
    """svc_inference.py:25-28 default: identity (the reference clones to CPU; the engine keeps the
    chunk on the device since nothing is changed)."""

    def retriv_whisper(self, vec):
        return vec

    def retriv_hubert(self, vec):
        return vec


def load_svc_model(checkpoint_path, model):
    """svc_inference.py:61-74: tolerant load of ``ckpt["model_g"]`` -- keys absent from the checkpoint keep
    the model's current value and are reported, exactly like the reference."""
    assert os.path.isfile(checkpoint_path)
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu")
    saved = checkpoint_dict["model_g"]
    state = model.state_dict()
    new_state = {}
    for k, v in state.items():
        if k in saved:
            new_state[k] = saved[k]
        else:
            print("%s is not in the checkpoint" % k)
            new_state[k] = v
    model.load_state_dict(new_state)
    return model


def chunk_schedule(all_frame, hop_size, out_chunk=K.CHUNK_FRAMES, hop_frame=K.HALO_FRAMES):
    """(cut_s, cut_e, cut_s_out, cut_e_out) per chunk -- the arithmetic of svc_inference.py:101-115."""
    plan, out_index = [], 0
    while out_index < all_frame:
        if out_index == 0:
            cut_s, cut_s_out = 0, 0
        else:
            cut_s, cut_s_out = out_index - hop_frame, hop_frame * hop_size
        if out_index + out_chunk + hop_frame > all_frame:
            cut_e, cut_e_out = all_frame, -1
        else:
            cut_e, cut_e_out = out_index + out_chunk + hop_frame, -1 * hop_frame * hop_size
        plan.append((cut_s, cut_e, cut_s_out, cut_e_out))
        out_index += out_chunk
    return plan


def _chunk_streams(model, dev, n):
    """``n`` side streams cached on the model (none for n <= 1)."""
    if n <= 1:
        return []
    pools = model.__dict__.setdefault("_svcmi_chunk_streams", {})      # per calling thread: concurrent conversions do not share side streams
    pool = pools.setdefault(threading.get_ident(), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


@torch.no_grad()
def svc_infer(model, retrieval, spk, pit, ppg, vec, hp, device, noise=None, write_pit_wav=True, return_tensor=False):
    """svc_inference.py:77-134.  spk [spk_dim], pit [T] Hz, ppg [T, ppg_dim], vec [T, vec_dim] (already
    repeated x2 to 100 fps, :175-182).  ``noise`` (optional, for reproducible runs) =
    {"rand_ini": [1,11], "src_noise": [1,L,11], "enc_noises": [[1,inter,len_i] per chunk]}.
    Returns np.float32 [hop*T - 1] (or the device tensor with ``return_tensor``)."""
    dev = torch.device(device)
    n = min(pit.shape[0], vec.shape[0], ppg.shape[0])
    pit = pit[:n].to(dev, torch.float32)
    vec = vec[:n].to(dev, torch.float32)
    ppg = ppg[:n].to(dev, torch.float32)
    spk = spk.to(dev, torch.float32).unsqueeze(0)
    src_noise = None if noise is None else (noise["rand_ini"], noise["src_noise"])
    source = model.pitch2source(pit.unsqueeze(0), noise=src_noise)
    if write_pit_wav:
        from scipy.io.wavfile import write
        write("svc_out_pit.wav", hp.data.sampling_rate, model.source2wav(source))
    hop = hp.data.hop_length
    retrieval = retrieval if retrieval is not None else DummyRetrieval()
    passthrough = isinstance(retrieval, DummyRetrieval)
    pieces = []
    plan = chunk_schedule(n, hop)
    # Chunks are independent once the source is made: with ``model.chunk_streams = N`` > 1 they are issued round-robin on N HIP streams
    # (clips in flight, svcmi/lanes.py: one chunk's latency-bound launches run beside another chunk's GEMMs); same launches, same
    # results.  User retrieval hooks work on CPU tensors and keep the serial order.
    side = []
    if dev.type == "cuda" and len(plan) > 1 and (passthrough or getattr(retrieval, "on_device", False)):
        side = _chunk_streams(model, dev, min(int(getattr(model, "chunk_streams", 1) or 1), len(plan)))
    main_stream = torch.cuda.current_stream(dev) if side else None
    for s in side:
        s.wait_stream(main_stream)
    for i, (cs, ce, cso, ceo) in enumerate(plan):
        with (torch.cuda.stream(side[i % len(side)]) if side else contextlib.nullcontext()):
            sub_ppg, sub_vec = ppg[cs:ce], vec[cs:ce]
            if getattr(retrieval, "on_device", False):      # svcmi.feature_retrieval.KnnIndexRetrieval: the GPU kNN blend
                sub_ppg, sub_vec = retrieval.retriv_whisper(sub_ppg), retrieval.retriv_hubert(sub_vec)
            elif not passthrough:      # user hook works on CPU tensors (feature_retrieval/retrieval.py:11-28)
                sub_ppg = retrieval.retriv_whisper(sub_ppg.cpu()).to(dev)
                sub_vec = retrieval.retriv_hubert(sub_vec.cpu()).to(dev)
            sub_len = torch.tensor([ce - cs], dtype=torch.int64)
            enc_noise = None if noise is None else noise["enc_noises"][i]
            sub_out = model.inference(sub_ppg.unsqueeze(0), sub_vec.unsqueeze(0), pit[cs:ce].unsqueeze(0), spk, sub_len,
                                      source[:, :, cs * hop:ce * hop], noise=enc_noise)
            pieces.append(sub_out[0, 0, cso:ceo])
    for s in side:
        main_stream.wait_stream(s)
    for p in pieces if side else ():
        p.record_stream(main_stream)              # allocated on a side stream, read by the concatenation below
    out = torch.cat(pieces)
    return out if return_tensor else out.cpu().numpy()


@torch.no_grad()
def extract_features(audio, whisper, hubert, crepe, device, in_flight=True):
    """The three extractors of svc_inference.py:138-154 on ONE 16 kHz waveform (numpy float32 [n]), in process and in flight together:
    the reference runs them as three child processes one after the other; they are independent, so here the Whisper PPG, the HuBERT
    units and the CREPE F0 track are issued on three HIP streams (clips-in-flight idea of svcmi/lanes.py: one network's launch gaps
    are filled by another's kernels).  Returns (ppg [T50, ppg_dim] device, vec [T50, vec_dim] device, f0 np.float32 [2 * (1 + n // 320)])
    -- the contents of svc_tmp.ppg.npy / svc_tmp.vec.npy / svc_tmp.pit.csv (before the CSV's int() quantisation)."""
    from .hubert import inference as hubert_inf
    from .pitch import inference as pitch_inf
    from .whisper import inference as whisper_inf
    dev = torch.device(device)
    if dev.type != "cuda" or not in_flight:
        return (whisper_inf.ppg_from_audio(whisper, audio), hubert_inf.units_windowed(hubert, audio),
                pitch_inf.compute_f0_sing(audio, dev, model=crepe))
    cur = torch.cuda.current_stream(dev)
    pool = whisper.__dict__.setdefault("_svcmi_extract_streams", {})          # per calling thread (folder driver workers)
    side = pool.setdefault(threading.get_ident(), [])
    while len(side) < 2:
        side.append(torch.cuda.Stream(device=dev))
    for s in side:
        s.wait_stream(cur)
    f0_finish = pitch_inf.compute_f0_sing_begin(audio, dev, model=crepe)       # the longest of the three first, on the current stream
    with torch.cuda.stream(side[0]):
        ppg = whisper_inf.ppg_from_audio(whisper, audio)
    with torch.cuda.stream(side[1]):
        vec = hubert_inf.units_windowed(hubert, audio)
    f0 = f0_finish()                                                          # D2H copy of the decoded track + host tail
    for s, t in zip(side, (ppg, vec)):
        cur.wait_stream(s)
        t.record_stream(cur)
    return ppg, vec, f0


# ------------------------------------------------------------------------------------------------ CLI (svc_inference.py:137-239)
class _Hp(dict):
    """Attribute-style view of the YAML config, like the OmegaConf object the reference passes around
    (``hp.vits.ppg_dim`` ...); missing attributes raise AttributeError so hasattr / deepcopy behave."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return _Hp(v) if isinstance(v, dict) else v


def load_config(path):
    """``OmegaConf.load(args.config)`` for configs/base.yaml-style files (plain YAML; OmegaConf is not required)."""
    import yaml
    with open(path, "r", encoding="utf-8") as f:
        return _Hp(yaml.safe_load(f))


def shift_pitch(pit, shift):
    """svc_inference.py:185-200: transpose the F0 track by ``shift`` semitones (0 = untouched)."""
    pit = np.asarray(pit)
    if shift == 0:
        return pit
    source = pit[pit > 0]
    print(f"source pitch statics: mean={source.mean():0.1f}, min={source.min():0.1f}, max={source.max():0.1f}")
    return pit * 2 ** (shift / 12)


def main(args):
    """The reference's ``main`` with the three feature extractors run IN PROCESS on the GPU instead of as
    ``os.system("python whisper|hubert|pitch/inference.py ...")`` children that each reload their model (:138-154).
    The intermediate files keep their names and formats (svc_tmp.ppg.npy / .vec.npy / .pit.csv), so ``--ppg/--vec/--pit``
    work as before; ``svc_out.wav`` (float32, hp.data.sampling_rate) and ``svc_out_pit.wav`` are written like the
    reference does."""
    from scipy.io.wavfile import write
    from .hubert import inference as hubert_inf
    from .pitch import inference as pitch_inf
    from .vits.models import SynthesizerInfer
    from .whisper import inference as whisper_inf
    device = "cuda"
    f0_prec = None if getattr(args, "f0_precision", "f32") == "f32" else getattr(args, "f0_precision", "f32")
    prec = None if getattr(args, "precision", "f32") == "f32" else args.precision

    enc_prec = "f16" if prec == "mixed" else prec      # the mixed policy's classes are the synthesizer's; the extractors follow the reference's .half()

    def _with_prec(m):
        (m.encoder if hasattr(m, "encoder") else m).precision = enc_prec
        return m
    if args.ppg is None and args.vec is None and args.pit is None:
        # all three features from the wav: the extractors run in flight together (extract_features); the intermediate files keep
        # their names and formats
        from .whisper.audio import load_audio
        crepe = pitch_inf.load_crepe(args.crepe, device)
        crepe.precision = f0_prec
        ppg_d, vec_d, f0 = extract_features(load_audio(args.wave), _with_prec(whisper_inf.load_model(args.whisper, device)),
                                            _with_prec(hubert_inf.load_model(args.hubert, device)), crepe, device)
        args.ppg, args.vec, args.pit = "svc_tmp.ppg.npy", "svc_tmp.vec.npy", "svc_tmp.pit.csv"
        np.save(args.ppg, ppg_d.cpu().numpy(), allow_pickle=False)
        np.save(args.vec, vec_d.cpu().numpy(), allow_pickle=False)
        pitch_inf.save_csv_pitch(f0, args.pit)
    if args.ppg is None:
        args.ppg = "svc_tmp.ppg.npy"
        whisper_inf.pred_ppg(_with_prec(whisper_inf.load_model(args.whisper, device)), args.wave, args.ppg, device)
    if args.vec is None:
        args.vec = "svc_tmp.vec.npy"
        hubert_inf.pred_vec(_with_prec(hubert_inf.load_model(args.hubert, device)), args.wave, args.vec, device)
    if args.pit is None:
        args.pit = "svc_tmp.pit.csv"
        crepe = pitch_inf.load_crepe(args.crepe, device)
        crepe.precision = f0_prec
        pitch_inf.save_csv_pitch(pitch_inf.compute_f0_sing(args.wave, device, model=crepe), args.pit)
    hp = load_config(args.config)
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp)
    load_svc_model(args.model, model)
    from .feature_retrieval import create_retrival
    retrieval = create_retrival(args, device)      # DummyRetrieval unless --enable-retrieval (:25-58)
    model.eval()
    model.to(device)
    model.precision = prec
    spk = torch.FloatTensor(np.load(args.spk))
    ppg = torch.FloatTensor(np.repeat(np.load(args.ppg), 2, 0))      # 320 PPG -> 160 * 2 (:175-177)
    vec = torch.FloatTensor(np.repeat(np.load(args.vec), 2, 0))
    print("pitch shift: ", args.shift)
    pit = torch.FloatTensor(shift_pitch(pitch_inf.load_csv_pitch(args.pit), args.shift))
    out_audio = svc_infer(model, retrieval, spk, pit, ppg, vec, hp, device)
    write("svc_out.wav", hp.data.sampling_rate, out_audio)
    return out_audio


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="svcmi drop-in for the reference's svc_inference.py")
    p.add_argument("--config", type=str, required=True, help="yaml file for config.")
    p.add_argument("--model", type=str, required=True, help="path of model for evaluation")
    p.add_argument("--wave", type=str, required=True, help="Path of raw audio.")
    p.add_argument("--spk", type=str, required=True, help="Path of speaker.")
    p.add_argument("--ppg", type=str, help="Path of content vector.")
    p.add_argument("--vec", type=str, help="Path of hubert vector.")
    p.add_argument("--pit", type=str, help="Path of pitch csv file.")
    p.add_argument("--shift", type=int, default=0, help="Pitch shift key.")
    p.add_argument("--enable-retrieval", action="store_true", help="Enable index feature retrieval")
    p.add_argument("--retrieval-index-prefix", default="",
                   help="retrieval index file prefix. Will load file %%prefix%%hubert.index/%%prefix%%whisper.index")
    p.add_argument("--retrieval-ratio", type=float, default=.5, help="ratio of feature retrieval effect. Must be in range 0..1")
    p.add_argument("--n-retrieval-vectors", type=int, default=3, choices=range(1, 9), metavar="[1-8]",
                   help="get n nearest vectors from retrieval index (1..8)")
    p.add_argument("--hubert-index-path", required=False, help="path to a hubert IVF-Flat .index (or a .npy [n, 256] bank: exact search)")
    p.add_argument("--whisper-index-path", required=False, help="path to a whisper IVF-Flat .index (or a .npy [n, 1280] bank: exact search)")
    p.add_argument("--whisper", type=str, default=os.path.join("whisper_pretrain", "large-v2.pt"))
    p.add_argument("--hubert", type=str, default=os.path.join("hubert_pretrain", "hubert-soft-0d54a1f4.pt"))
    p.add_argument("--crepe", type=str, default=os.path.join("crepe", "assets", "full.pth"))
    p.add_argument("--debug", action="store_true")
    p.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "f16", "mixed"],
                   help="GEMM operand precision of the Whisper / HuBERT encoders and the synthesizer (fp32 accumulation, LayerNorm / softmax / "
                        "SnakeAlias in fp32 in every mode): f32 = the reference CPU path's arithmetic (parity default); bf16x3 = split-bf16, "
                        "waveform within 2e-5 of fp32; f16 = what the reference's .half() accelerator path does (waveform within 1e-3); bf16 (7e-3)")
    p.add_argument("--f0-precision", default="f32", choices=["f32", "bf16x3", "f16", "bf16"],
                   help="GEMM operand precision of the CREPE F0 extractor: bf16x3 = split-bf16 operands, fp32 accumulate (posteriors within 1e-5 "
                        "of fp32, decoded track identical on the golden clip: tests/test_gpu_engine.py), 2x faster; f32 = exact-fp32 matrix cores; "
                        "f16 = fp16 operands and activations (posteriors within 5e-4, decoded track identical on the test clip: "
                        "tests/test_gpu_precision.py; what an accelerator run of the reference is closest to), another 2x faster; bf16 (4e-3)")
    return p


if __name__ == "__main__":
    from svcmi.lanes import want_hw_queues
    want_hw_queues()                      # before the first HIP call: the chunk streams of svc_infer get their own hardware queues
    main(build_parser().parse_args())
