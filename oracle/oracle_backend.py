"""TEST INFRASTRUCTURE — `oracle.whisper.OracleWhisper` behind the `ctranslate2.models.Whisper` interface.

Lets the SAME host code (faster_whisper_amd.transcribe: batched pipeline, sequential seek loop, temperature
fallback, word timestamps) run once on the HIP engine and once on the CPU restatement, so the GPU tests compare
whole transcriptions and not only single backend calls.  CPU only, slow, never imported by the product.
"""
import itertools
from typing import List, Sequence

import numpy as np

from . import logmel as olm
from .whisper import OracleWhisper


class _Enc:
    def __init__(self, array: np.ndarray):
        self.array = array
        self.shape = list(array.shape)

    def to_numpy(self):
        return self.array


class OracleBackend:
    def __init__(self, cfg, weights, emulate_fp16: bool = True, int8: bool = False):
        from faster_whisper_amd.backend import language_token_strings
        self.config = cfg
        self.oracle = OracleWhisper(cfg, weights, emulate_fp16=emulate_fp16, int8=int8)
        self.is_multilingual = cfg.is_multilingual
        self.n_mels = cfg.n_mels
        self.device, self.device_index = "cpu", [0]
        self._lang_names = language_token_strings(cfg)
        self._seed_counter = itertools.count(1)   # mirrors backend.Whisper: one seed per sampling call

    def log_mel_full(self, pcm):
        return olm.log_mel_full(np.asarray(pcm, dtype=np.float32), self.n_mels)

    def log_mel(self, chunks: Sequence[np.ndarray]):
        return olm.log_mel_chunks(list(chunks), self.n_mels)

    def encode_pcm(self, chunks):
        return self.encode(self.log_mel(chunks))

    def encode(self, features, to_cpu: bool = False):
        f = np.asarray(features, dtype=np.float32)
        if f.ndim == 2:
            f = f[None]
        return _Enc(self.oracle.encode(f))

    def generate(self, enc, prompts: List[List[int]], **kw):
        from faster_whisper_amd.backend import call_seed
        kw.pop("asynchronous", None)
        if kw.get("seed") is None:
            kw["seed"] = call_seed(next(self._seed_counter))
        return self.oracle.generate(enc.array, prompts, **kw)

    def detect_language(self, enc):
        c = self.config
        return [[(self._lang_names[tok - c.lang_begin], p) for tok, p in row]
                for row in self.oracle.detect_language(enc.array)]

    def align(self, enc, start_sequence, text_tokens, num_frames, *, median_filter_width: int = 7):
        return self.oracle.align(enc.array, start_sequence, text_tokens, num_frames,
                                 median_filter_width=median_filter_width)
