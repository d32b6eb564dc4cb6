"""Hyper-parameters the reference hard-codes in Python rather than in configs/base.yaml
(SURVEY.md section 5 "Config / flags"); the engine takes the same YAML plus these."""
ENC_HEADS = 2          # vits/models.py:226
ENC_LAYERS = 6         # vits/models.py:227
ENC_FFN_KERNEL = 3     # vits/models.py:228
ENC_WINDOW = 4         # vits/attentions.py:21
FLOW_KERNEL = 5        # vits/models.py:234
FLOW_WN_LAYERS = 4     # vits/models.py:236
FLOW_N = 4             # vits/models.py:63
NSF_HARMONICS = 11     # vits_decoder/nsf.py:361 (harmonic_num 10 + fundamental)
CHUNK_FRAMES = 2500    # svc_inference.py:97
HALO_FRAMES = 10       # svc_inference.py:96
WHISPER_WINDOW_S = 15  # whisper/inference.py:37
