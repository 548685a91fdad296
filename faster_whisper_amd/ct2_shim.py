"""Module-compatible stand-in for `ctranslate2`, so the REFERENCE host code
(faster_whisper/transcribe.py, unmodified) runs on the MI355X engine:

    import faster_whisper_amd.ct2_shim as shim
    shim.install()                    # registers `ctranslate2` (+ `ctranslate2.models`) in sys.modules
    import faster_whisper             # the reference package
    model = faster_whisper.WhisperModel("/path/to/fwamd_model_dir", device="cuda")

Only what the reference touches is provided: `models.Whisper`,
`models.WhisperGenerationResult`, `models.WhisperAlignmentResult`, `StorageView`.
"""
import sys
import types

from .backend import StorageView, Whisper, WhisperAlignmentResult, WhisperGenerationResult

__version__ = "4.5.0+fwamd"

models = types.ModuleType("ctranslate2.models")
models.Whisper = Whisper
models.WhisperGenerationResult = WhisperGenerationResult
models.WhisperAlignmentResult = WhisperAlignmentResult


def get_supported_compute_types(device: str, device_index: int = 0):
    if device not in ("cuda", "auto"):
        return set()
    return {"float16", "int8_float16"}


def get_cuda_device_count() -> int:
    from . import _lib
    return int(_lib.load().fw_device_count())


def install(stub_av: bool = False):
    """Register this module as `ctranslate2`.  stub_av=True also registers an empty `av` module
    (PyAV is only needed by decode_audio(); ndarray inputs never reach it)."""
    mod = sys.modules[__name__]
    sys.modules["ctranslate2"] = mod
    sys.modules["ctranslate2.models"] = models
    if stub_av and "av" not in sys.modules:
        av = types.ModuleType("av")
        av.error = types.SimpleNamespace(InvalidDataError=Exception)
        sys.modules["av"] = av
    return mod
