"""svcmi.hubert -- HuBERT-Soft content units on the svcmi kernels (drop-in for the reference's hubert/ package)."""
from .inference import HubertSoft, load_model, pred_vec  # noqa: F401
