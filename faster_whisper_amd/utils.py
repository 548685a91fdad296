"""Small public helpers of the reference's top level (faster_whisper/utils.py): `format_timestamp` (:119-139) and
`available_models` (:11-31, :34-36).  Downloading converted models from the Hugging Face Hub (`download_model`) is
outside this tier: pass a local CTranslate2 model directory to WhisperModel."""
from typing import List

# model sizes the reference resolves to hub repositories; the geometry of each is in config.get_config
_MODEL_SIZES = ("tiny.en", "tiny", "base.en", "base", "small.en", "small", "medium.en", "medium", "large-v1",
                "large-v2", "large-v3", "large", "distil-large-v2", "distil-medium.en", "distil-small.en",
                "distil-large-v3", "distil-large-v3.5", "large-v3-turbo", "turbo")


def available_models() -> List[str]:
    """names of the model sizes the reference knows"""
    return list(_MODEL_SIZES)


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = ".") -> str:
    """[HH:]MM:SS.mmm (hours only when non-zero unless forced)"""
    assert seconds >= 0, "non-negative timestamp expected"
    ms = round(seconds * 1000.0)
    hours, ms = divmod(ms, 3_600_000)
    minutes, ms = divmod(ms, 60_000)
    secs, ms = divmod(ms, 1_000)
    prefix = f"{hours:02d}:" if always_include_hours or hours > 0 else ""
    return f"{prefix}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"
