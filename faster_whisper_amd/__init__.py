"""faster_whisper_amd — MI355X-native (gfx950, HIP) Whisper transcription engine that keeps
faster-whisper's `WhisperModel` / `BatchedInferencePipeline.transcribe()` surface and replaces
the `ctranslate2.models.Whisper` backend (see DESIGN.md / INTEGRATION.md)."""
from .backend import (StorageView, Whisper, WhisperAlignmentResult, WhisperGenerationResult, pack_blob,
                      save_model_dir)
from .config import WhisperConfig, get_config
from .weights import synthetic_weights

__version__ = "0.1.0"

__all__ = [
    "Whisper", "StorageView", "WhisperGenerationResult", "WhisperAlignmentResult", "WhisperConfig",
    "get_config", "synthetic_weights", "pack_blob", "save_model_dir",
]


def __getattr__(name):
    # the host pipeline pulls in tokenizers/tqdm; import it lazily
    if name in ("WhisperModel", "BatchedInferencePipeline", "Segment", "Word", "TranscriptionInfo",
                "TranscriptionOptions", "FeatureExtractor", "Tokenizer"):
        from . import transcribe as _t
        return getattr(_t, name)
    # the rest of the reference's top level (faster_whisper/__init__.py)
    if name == "decode_audio":
        from .audio import decode_audio
        return decode_audio
    if name in ("format_timestamp", "available_models"):
        from . import utils as _u
        return getattr(_u, name)
    raise AttributeError(name)
