"""TEST INFRASTRUCTURE — a deterministic stand-in for `ctranslate2.models.Whisper`.

Every output is a pure function of the call's inputs (a coarse fingerprint of the encoder input, the prompt,
the decoding arguments), so the REFERENCE's host code (oracle/gen_golden_host.py, run in the build container)
and this repository's host code (tests/test_host_golden.py) see the same "model" and must produce the same
segments.  The scripts are chosen to reach the branches of the host logic: consecutive / single-ended / no
timestamps, repetitive text (compression-ratio fallback), improbable text (log-prob fallback), silence
(no-speech skip), word alignments with punctuation and multi-byte characters.

Not a model: nothing here is used by the product path.  Only tests/ and oracle/ import this module.
"""
import zlib
from typing import List, Sequence

import numpy as np

from . import logmel as olm

_TEXT = (" hello world, this is a test. the model was not ( ready ) for \"speech\" and time? we have one word!"
         " 世界 café you and I: they [ whisper ] on audio - stamp that it is for testing the words")


class Result:
    def __init__(self, sequences_ids, scores, no_speech_prob):
        self.sequences_ids = sequences_ids
        self.scores = scores
        self.no_speech_prob = no_speech_prob


class Alignment:
    def __init__(self, alignments, text_token_probs):
        self.alignments = alignments
        self.text_token_probs = text_token_probs


class EncoderOutput:
    def __init__(self, fingerprints: List[int]):
        self.fp = fingerprints
        self.shape = [len(fingerprints), 1500, 8]


def fingerprint(feat: np.ndarray) -> int:
    """coarse, rounding-tolerant signature of one [n_mels, 3000] feature image: the mean level of 12 time
    bins to 1 decimal"""
    f = np.asarray(feat, dtype=np.float64)
    bins = f.reshape(f.shape[0], 12, -1).mean(axis=(0, 2))
    key = ",".join(f"{v:.1f}" for v in bins)
    return zlib.crc32(key.encode()) & 0x7FFFFFFF


class ScriptedBackend:
    def __init__(self, cfg, hf_tokenizer, silence_level: float = -0.45):
        """cfg: faster_whisper_amd WhisperConfig-like (token ids + n_mels); hf_tokenizer: the micro tokenizer"""
        self.config = cfg
        self.is_multilingual = cfg.is_multilingual
        self.n_mels = cfg.n_mels
        self.device = "cpu"
        self.device_index = [0]
        self.silence_level = silence_level
        self.pool = hf_tokenizer.encode(_TEXT, add_special_tokens=False).ids
        self.calls = []            # (kind, details) log for the tests

    # ---- features (this repository's FeatureExtractor / fused path hooks) ----------------
    def log_mel_full(self, pcm: np.ndarray) -> np.ndarray:
        return olm.log_mel_full(np.asarray(pcm, dtype=np.float32), self.n_mels)

    def log_mel(self, chunks: Sequence[np.ndarray]) -> np.ndarray:
        return olm.log_mel_chunks(list(chunks), self.n_mels)

    def encode_pcm(self, chunks: Sequence[np.ndarray]):
        return self.encode(self.log_mel(chunks))

    # ---- ctranslate2.models.Whisper interface ----------------------------------------------
    def encode(self, features, to_cpu: bool = False):
        f = np.asarray(features)
        if f.ndim == 2:
            f = f[None]
        assert f.shape[1] == self.n_mels and f.shape[2] == 3000, f.shape
        self._last_level = [float(x.mean()) for x in f]
        out = EncoderOutput([fingerprint(x) for x in f])
        out.level = self._last_level
        self.calls.append(("encode", out.fp))
        return out

    def detect_language(self, enc):
        names = ["<|en|>", "<|zh|>", "<|de|>", "<|es|>"][: max(1, self.config.n_langs)]
        out = []
        for fp in enc.fp:
            rng = np.random.default_rng([fp, 77])
            p = rng.dirichlet(np.ones(len(names)) * 0.4)
            order = np.argsort(-p, kind="stable")
            out.append([(names[i], float(p[i])) for i in order])
        self.calls.append(("detect_language", enc.fp))
        return out

    def generate(self, enc, prompts, *, beam_size=5, patience=1, num_hypotheses=1, length_penalty=1,
                 repetition_penalty=1, no_repeat_ngram_size=0, max_length=448, return_scores=False,
                 return_no_speech_prob=False, max_initial_timestamp_index=50, suppress_blank=True,
                 suppress_tokens=None, sampling_topk=1, sampling_temperature=1, **_):
        cfg = self.config
        tb = cfg.timestamp_begin
        out = []
        for row, (fp, prompt) in enumerate(zip(enc.fp, prompts)):
            sampling = beam_size == 1 and num_hypotheses > 1
            t10 = int(round(sampling_temperature * 10)) if sampling else 0
            rng = np.random.default_rng([fp, t10, len(prompt) % 5])
            budget = max(1, max_length - len(prompt))
            with_ts = not (len(prompt) > 0 and prompt[-1] == cfg.no_timestamps)
            silent = enc.level[row] < self.silence_level
            u = rng.random()
            if silent:
                no_speech = float(rng.choice([0.75, 0.95]))
                avg = float(rng.uniform(-1.8, -1.1))
            else:
                no_speech = float(rng.choice([0.01, 0.2, 0.65], p=[0.6, 0.3, 0.1]))
                avg = float(rng.uniform(-1.4, -0.95) if u < (0.35 if t10 < 4 else 0.1) else rng.uniform(-0.8, -0.1))
            repetitive = (not silent) and rng.random() < (0.4 if t10 == 0 else 0.25 if t10 <= 2 else 0.0)

            def text(n):
                if repetitive:
                    base = [self.pool[int(rng.integers(len(self.pool)))] for _ in range(2)]
                    return (base * n)[:n]
                start = int(rng.integers(len(self.pool)))
                return [self.pool[(start + i) % len(self.pool)] for i in range(n)]

            if not with_ts:
                toks = text(int(rng.integers(6, 40)))
            else:
                layout = int(rng.integers(5))
                t0 = int(rng.integers(0, 1 + min(60, max_initial_timestamp_index + 10)))
                if layout == 0:      # two closed segments, ends on a lone timestamp
                    t1 = t0 + int(rng.integers(100, 500))
                    t2 = t1 + int(rng.integers(100, 500))
                    toks = [tb + t0] + text(int(rng.integers(4, 30))) + [tb + t1, tb + t1] + \
                        text(int(rng.integers(4, 30))) + [tb + min(t2, 1500)]
                elif layout == 1:    # closed segment + open tail (no final timestamp)
                    t1 = t0 + int(rng.integers(150, 700))
                    toks = [tb + t0] + text(int(rng.integers(4, 30))) + [tb + t1, tb + t1] + \
                        text(int(rng.integers(3, 20)))
                elif layout == 2:    # one segment closed by consecutive timestamps at the very end
                    t1 = t0 + int(rng.integers(300, 1300))
                    toks = [tb + t0] + text(int(rng.integers(8, 50))) + [tb + min(t1, 1500), tb + min(t1, 1500)]
                elif layout == 3:    # single segment with a lone final timestamp
                    t1 = t0 + int(rng.integers(200, 1400))
                    toks = [tb + t0] + text(int(rng.integers(5, 40))) + [tb + min(t1, 1500)]
                else:                # no timestamps at all
                    toks = text(int(rng.integers(5, 40)))
            if repetitive:
                toks = (toks * 8)
            toks = toks[:budget]
            n = len(toks)
            score = avg * (n + 1) / (n ** length_penalty)
            out.append(Result([list(map(int, toks))], [float(score)], no_speech))
        self.calls.append(("generate", dict(beam_size=beam_size, num_hypotheses=num_hypotheses,
                                            temperature=float(sampling_temperature), max_length=max_length,
                                            prompts=[[int(t) for t in p] for p in prompts],
                                            patience=float(patience), length_penalty=float(length_penalty),
                                            repetition_penalty=float(repetition_penalty),
                                            no_repeat_ngram_size=int(no_repeat_ngram_size),
                                            max_initial_timestamp_index=int(max_initial_timestamp_index),
                                            suppress_blank=bool(suppress_blank),
                                            suppress_tokens=(None if suppress_tokens is None
                                                             else [int(t) for t in suppress_tokens]))))
        return out

    def align(self, enc, start_sequence, text_tokens, num_frames, *, median_filter_width=7):
        out = []
        frames = num_frames if isinstance(num_frames, (list, tuple)) else [num_frames] * len(text_tokens)
        fps = enc.fp if len(enc.fp) == len(text_tokens) else [enc.fp[0]] * len(text_tokens)
        for fp, toks, nf in zip(fps, text_tokens, frames):
            rng = np.random.default_rng([fp, len(toks), 5])
            T = len(toks) + 1
            F = max(1, int(nf) // 2)
            # token j starts at frame b[j]; monotone, jittered, the path visits every token at least once
            b = np.floor(np.arange(T) * F / T + rng.uniform(0, 0.9 * F / T, size=T)).astype(int)
            b = np.minimum(np.maximum.accumulate(b), F - 1)
            b[0] = 0
            pairs = []
            for j in range(T):
                end = (b[j + 1] - 1) if j + 1 < T else F - 1
                for t in range(int(b[j]), int(max(b[j], end)) + 1):
                    pairs.append((j, t))
            probs = [float(x) for x in rng.uniform(0.02, 1.0, size=len(toks))]
            out.append(Alignment(pairs, probs))
        self.calls.append(("align", dict(n=[len(t) for t in text_tokens], frames=[int(f) for f in frames])))
        return out
