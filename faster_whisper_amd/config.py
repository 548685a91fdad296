"""Whisper geometry + vocabulary layout (what CTranslate2 reads from model.bin/config.json).

Token ids: SURVEY.md Appendix A.2; the `.en` row is pinned by the reference's
tests/test_tokenizer.py:110, `timestamp_begin = no_timestamps + 1` by tokenizer.py:76-78.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

from . import _lib


@dataclass
class WhisperConfig:
    name: str
    n_mels: int
    d_model: int
    n_heads: int
    n_enc_layers: int
    n_dec_layers: int
    n_vocab: int
    is_multilingual: bool
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448
    eot: int = 0
    sot: int = 0
    lang_begin: int = 0
    n_langs: int = 0
    translate: int = 0
    transcribe: int = 0
    sot_lm: int = 0
    sot_prev: int = 0
    no_speech: int = 0
    no_timestamps: int = 0
    timestamp_begin: int = 0
    suppress_begin: Tuple[int, ...] = ()
    alignment_heads: List[Tuple[int, int]] = field(default_factory=list)

    def to_c(self) -> "_lib.FwConfig":
        c = _lib.FwConfig()
        c.n_mels, c.n_audio_ctx, c.d_model, c.n_heads = self.n_mels, self.n_audio_ctx, self.d_model, self.n_heads
        c.n_enc_layers, c.n_dec_layers, c.n_vocab, c.n_text_ctx = (
            self.n_enc_layers, self.n_dec_layers, self.n_vocab, self.n_text_ctx)
        c.is_multilingual = int(self.is_multilingual)
        c.tok_eot, c.tok_sot, c.tok_lang_begin, c.n_langs = self.eot, self.sot, self.lang_begin, self.n_langs
        c.tok_translate, c.tok_transcribe, c.tok_sot_lm, c.tok_sot_prev = (
            self.translate, self.transcribe, self.sot_lm, self.sot_prev)
        c.tok_no_speech, c.tok_no_timestamps, c.tok_timestamp_begin = (
            self.no_speech, self.no_timestamps, self.timestamp_begin)
        c.n_suppress_begin = len(self.suppress_begin)
        for i, t in enumerate(self.suppress_begin):
            c.suppress_begin[i] = t
        c.n_align_heads = len(self.alignment_heads)
        for i, (l, h) in enumerate(self.alignment_heads):
            c.align_heads[2 * i], c.align_heads[2 * i + 1] = l, h
        return c

    @property
    def sot_sequence(self) -> List[int]:
        """[sot] for English-only models, [sot, <|en|>, <|transcribe|>] otherwise (tokenizer.py:37-40,81-90)."""
        if not self.is_multilingual:
            return [self.sot]
        return [self.sot, self.lang_begin, self.transcribe]


def _vocab_en():   # gpt2 vocabulary (*.en models)
    return dict(n_vocab=51864, is_multilingual=False, eot=50256, sot=50257, lang_begin=50258, n_langs=0,
                translate=50357, transcribe=50358, sot_lm=50359, sot_prev=50360, no_speech=50361,
                no_timestamps=50362, timestamp_begin=50363, suppress_begin=(220, 50256))


def _vocab_multi():  # multilingual <= large-v2 (99 languages)
    return dict(n_vocab=51865, is_multilingual=True, eot=50257, sot=50258, lang_begin=50259, n_langs=99,
                translate=50358, transcribe=50359, sot_lm=50360, sot_prev=50361, no_speech=50362,
                no_timestamps=50363, timestamp_begin=50364, suppress_begin=(220, 50257))


def _vocab_v3():  # large-v3 / distil-large-v3 / turbo (100 languages)
    return dict(n_vocab=51866, is_multilingual=True, eot=50257, sot=50258, lang_begin=50259, n_langs=100,
                translate=50359, transcribe=50360, sot_lm=50361, sot_prev=50362, no_speech=50363,
                no_timestamps=50364, timestamp_begin=50365, suppress_begin=(220, 50257))


def _vocab_micro():  # small synthetic vocabulary for fast tests: 400 text ids, 4 languages, 1501 timestamps
    return dict(n_vocab=412 + 1501, is_multilingual=True, eot=400, sot=401, lang_begin=402, n_langs=4,
                translate=406, transcribe=407, sot_lm=408, sot_prev=409, no_speech=410, no_timestamps=411,
                timestamp_begin=412, suppress_begin=(5, 400))


_DIMS = {
    "tiny": (384, 6, 4, 4), "base": (512, 8, 6, 6), "small": (768, 12, 12, 12), "medium": (1024, 16, 24, 24),
    "large": (1280, 20, 32, 32),
}


def get_config(name: str) -> WhisperConfig:
    """Geometry of the model sizes the reference can load (utils.py:11-31) plus the `micro` test model."""
    if name == "micro":
        return WhisperConfig(name=name, n_mels=80, d_model=128, n_heads=2, n_enc_layers=2, n_dec_layers=2,
                             **_vocab_micro())
    base = name
    en = base.endswith(".en")
    if en:
        base = base[:-3]
    if base in ("tiny", "base", "small", "medium"):
        d, h, le, ld = _DIMS[base]
        return WhisperConfig(name=name, n_mels=80, d_model=d, n_heads=h, n_enc_layers=le, n_dec_layers=ld,
                             **(_vocab_en() if en else _vocab_multi()))
    d, h, le, ld = _DIMS["large"]
    if name in ("large-v1", "large-v2", "distil-large-v2"):
        return WhisperConfig(name=name, n_mels=80, d_model=d, n_heads=h, n_enc_layers=le,
                             n_dec_layers=2 if name.startswith("distil") else ld, **_vocab_multi())
    if name in ("large-v3", "large"):
        return WhisperConfig(name=name, n_mels=128, d_model=d, n_heads=h, n_enc_layers=le, n_dec_layers=ld,
                             **_vocab_v3())
    if name in ("distil-large-v3", "distil-large-v3.5"):
        return WhisperConfig(name=name, n_mels=128, d_model=d, n_heads=h, n_enc_layers=le, n_dec_layers=2,
                             **_vocab_v3())
    if name in ("large-v3-turbo", "turbo"):
        return WhisperConfig(name=name, n_mels=128, d_model=d, n_heads=h, n_enc_layers=le, n_dec_layers=4,
                             **_vocab_v3())
    raise ValueError(f"unknown model geometry '{name}'")
