"""svcmi.pitch -- F0 extraction for the SVC pipeline (drop-in for the reference's pitch/inference.py + its vendored crepe)."""
from .inference import Crepe, bins_to_hz, compute_f0_sing, decode, load_csv_pitch, load_crepe, save_csv_pitch, viterbi_path  # noqa: F401
