"""Hyper-parameters of the hot path (workload definition shared by bench, tests and oracle).

The values restate /root/reference/configs/base.yaml:16-47 (data/vits/gen groups) and the
constants the reference hard-codes in Python instead of YAML ("Config"):
  vits/models.py:226-229   TextEncoder heads=2, layers=6, ffn kernel=3
  vits/models.py:234-237   flow kernel=5, dilation_rate=1, WN layers=4 ; models.py:63 n_flows=4
  vits/attentions.py:21    window_size=4
  vits_decoder/nsf.py:361-367  harmonic_num=10, sine_amp=0.1, noise_std=0.003
  vits_decoder/alias/act.py:112-115  2x up/down, 12 taps
  svc_inference.py:96-97   chunk 2500 frames, halo 10 frames
  whisper/inference.py:37  15 s windows ; whisper/inference.py:16-19 keep 24 of 32 blocks
"""


class AttrDict(dict):
    """yaml -> attribute tree; raises AttributeError (not KeyError) so copy/hasattr work."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v


BASE = {
    "data": {"sampling_rate": 32000, "hop_length": 320, "filter_length": 1024, "segment_size": 8000},
    "vits": {"ppg_dim": 1280, "vec_dim": 256, "spk_dim": 256, "gin_channels": 256,
             "inter_channels": 192, "hidden_channels": 192, "filter_channels": 640},
    "gen": {"upsample_input": 192, "upsample_rates": [5, 4, 4, 2, 2],
            "upsample_kernel_sizes": [15, 8, 8, 4, 4], "upsample_initial_channel": 320,
            "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]},
}


def base_hp():
    return AttrDict(BASE)


def tiny_hp():
    """A shrunken config with the same structure (every branch exercised) for fast CPU tests."""
    return AttrDict({
        "data": dict(BASE["data"]),
        "vits": {"ppg_dim": 64, "vec_dim": 32, "spk_dim": 32, "gin_channels": 32,
                 "inter_channels": 32, "hidden_channels": 32, "filter_channels": 48},
        "gen": {"upsample_input": 32, "upsample_rates": [5, 4, 4, 2, 2],
                "upsample_kernel_sizes": [15, 8, 8, 4, 4], "upsample_initial_channel": 64,
                "resblock_kernel_sizes": [3, 7, 11],
                "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]},
    })


# constants hard-coded in the reference's Python
ENC_HEADS = 2
ENC_LAYERS = 6
ENC_FFN_KERNEL = 3
ENC_WINDOW = 4
FLOW_KERNEL = 5
FLOW_WN_LAYERS = 4
FLOW_N = 4
NSF_HARMONICS = 11          # harmonic_num 10 + fundamental
NSF_SINE_AMP = 0.1
NSF_NOISE_STD = 0.003
# vits_decoder/nsf.py:378-381 fixed (non-trained) merge of the 11 harmonics
NSF_MERGE_W = [0.2942, -0.2243, 0.0033, -0.0056, -0.0020, -0.0046,
               0.0221, -0.0083, -0.0241, -0.0036, -0.0581]
NSF_MERGE_B = 0.0008
CHUNK_FRAMES = 2500
HALO_FRAMES = 10
WHISPER_WINDOW_S = 15

WHISPER_LARGE_V2 = {"n_mels": 80, "n_audio_ctx": 1500, "n_audio_state": 1280,
                    "n_audio_head": 20, "n_audio_layer": 32,
                    "n_vocab": 51865, "n_text_ctx": 448, "n_text_state": 1280,
                    "n_text_head": 20, "n_text_layer": 32}
WHISPER_TINY_TEST = {"n_mels": 80, "n_audio_ctx": 1500, "n_audio_state": 128,
                     "n_audio_head": 4, "n_audio_layer": 4,
                     "n_vocab": 64, "n_text_ctx": 8, "n_text_state": 128,
                     "n_text_head": 4, "n_text_layer": 1}


# HuBERT-Soft dimensions hard-coded in the reference (hubert/hubert_model.py:11-28,70-121) and a small set for the
# CPU-emulator tests (same structure, so every code path is exercised).
HUBERT_SOFT = {"conv_dim": 512, "embed": 768, "heads": 12, "ffn": 3072, "layers": 12, "pos_kernel": 128, "pos_groups": 16, "proj": 256}
HUBERT_TINY_TEST = {"conv_dim": 32, "embed": 64, "heads": 4, "ffn": 128, "layers": 2, "pos_kernel": 16, "pos_groups": 4, "proj": 16}
HUBERT_WINDOW_S = 20          # hubert/inference.py:30: 20 s windows
