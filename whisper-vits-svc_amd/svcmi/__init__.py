"""svcmi -- MI355X-native singing-voice-conversion inference (Whisper-PPG -> VITS flow -> NSF-BigVGAN).

Host-side mirror of the reference's Python surface over hand-written gfx950 kernels (libsvcmi.so,
include/svcmi.h).  Importing the package does not load the library; constructing ``Ops`` does, and raises
if the library or a GPU is missing -- there is no CPU fallback.
"""
from ._lib import SvcmiError, load_library  # noqa: F401
from .ops import Ops  # noqa: F401
from .svc_inference import DummyRetrieval, IRetrieval, chunk_schedule, load_svc_model, svc_infer  # noqa: F401
from .vits.models import SynthesizerInfer  # noqa: F401
from .whisper.inference import load_model as load_whisper_model  # noqa: F401

__version__ = "0.4.0"
