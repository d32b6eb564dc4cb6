"""Whisper content encoder on the svcmi kernels: log-mel front-end (audio.py) and the 24-block AudioEncoder (inference.py)."""
