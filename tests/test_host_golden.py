"""Host-logic parity PINNED against the reference: tests/golden/host_*.json were produced by running the
REFERENCE's own transcribe.py / tokenizer.py / vad.py (oracle/gen_golden_host.py, build container) on the
inputs of oracle/host_scenarios.py with oracle/scripted_backend.py as the model.  Here this repository's
host code runs on the same inputs with the same scripted model; segments, words, info and even the sequence
of backend calls must be identical.  No GPU."""
import dataclasses
import json
import logging
import os

import numpy as np
import pytest

from faster_whisper_amd import get_config
from faster_whisper_amd import vad as fvad
from faster_whisper_amd import words as fwords
from faster_whisper_amd.transcribe import BatchedInferencePipeline, FeatureExtractor, Tokenizer, WhisperModel
from oracle import host_scenarios as hs
from oracle import micro_tokenizer
from oracle.scripted_backend import ScriptedBackend

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "host_scenarios.json")) as f:
        scen = json.load(f)
    with open(os.path.join(GOLD, "host_units.json")) as f:
        units = json.load(f)
    return scen, units


@pytest.fixture(scope="module")
def hf_tok():
    return micro_tokenizer.build()


def make_model(cfg, hf_tok):
    m = WhisperModel.__new__(WhisperModel)
    m.logger = logging.getLogger("test")
    m.model = ScriptedBackend(cfg, hf_tok)
    m.hf_tokenizer = hf_tok
    m.feature_extractor = FeatureExtractor(feature_size=cfg.n_mels, backend=m.model)
    m.input_stride = 2
    m.num_samples_per_token = m.feature_extractor.hop_length * m.input_stride
    m.frames_per_second = m.feature_extractor.sampling_rate // m.feature_extractor.hop_length
    m.tokens_per_second = m.feature_extractor.sampling_rate // m.num_samples_per_token
    m.time_precision = 0.02
    m.max_length = 448
    return m


def _close(a, b, path=""):
    """structural equality with a 1e-9 float tolerance (json round trip of float64 is exact; the slack only
    absorbs np.float32-vs-float64 representation of probabilities)"""
    if isinstance(a, float) or isinstance(b, float):
        assert a is not None and b is not None, (path, a, b)
        assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b))), (path, a, b)
    elif isinstance(a, dict):
        assert isinstance(b, dict) and sorted(a) == sorted(b), (path, sorted(a), sorted(b) if isinstance(b, dict) else b)
        for k in a:
            _close(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert isinstance(b, (list, tuple)) and len(a) == len(b), (path, len(a), len(b) if hasattr(b, "__len__") else b)
        for i, (x, y) in enumerate(zip(a, b)):
            _close(x, y, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


def _plain(o):
    if dataclasses.is_dataclass(o):
        return {k: _plain(v) for k, v in dataclasses.asdict(o).items()}
    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    return o


@pytest.mark.parametrize("name", sorted(hs.SCENARIOS))
def test_transcribe_matches_reference_host_code(gold, hf_tok, name):
    scen, _ = gold
    sc = hs.SCENARIOS[name]
    cfg = get_config("micro")
    model = make_model(cfg, hf_tok)
    audio = hs.synth_audio(*sc["audio"])
    kwargs = json.loads(json.dumps(sc["kwargs"]))
    if kwargs.get("vad_filter"):
        # the Silero network is an input here (row f-3): same scripted probabilities the reference run used
        kwargs["vad_speech_probs"] = hs.speech_probs(np.pad(audio, (0, 512 - audio.shape[0] % 512)))
    if sc["kind"] == "sequential":
        segments, info = model.transcribe(audio, **kwargs)
    else:
        segments, info = BatchedInferencePipeline(model).transcribe(audio, **kwargs)
    segments = [_plain(s) for s in segments]
    want = scen[name]
    assert len(segments) == len(want["segments"]), (len(segments), len(want["segments"]))
    for i, (g, w) in enumerate(zip(segments, want["segments"])):
        _close(g, w, f"{name}.segments[{i}]")
    _close(dict(language=info.language, language_probability=float(info.language_probability),
                duration=info.duration, duration_after_vad=info.duration_after_vad,
                all_language_probs=_plain(info.all_language_probs)), want["info"], f"{name}.info")
    # same conversation with the backend: encode fingerprints, generate arguments per call, align sizes
    _close(_plain(model.model.calls), want["calls"], f"{name}.calls")


def test_vad_state_machine_chunks_and_time_map(gold):
    _, units = gold
    checked = 0
    for tname, (n_audio, probs) in hs.vad_prob_tracks().items():
        for cname, opts in hs.VAD_CASES.items():
            key = f"{tname}/{cname}"
            audio = np.zeros(n_audio, dtype=np.float32)
            spans = fvad.get_speech_timestamps(audio, fvad.VadOptions(**opts), speech_probs=probs)
            _close(_plain(spans), units["vad"][key], f"vad[{key}]")
            idx = np.arange(n_audio, dtype=np.float32)
            for md in (30.0, 7.5):
                chunks, meta = fvad.collect_chunks(idx, [dict(s) for s in spans], max_duration=md)
                got = dict(lens=[int(len(c)) for c in chunks], first=[float(c[0]) if len(c) else None for c in chunks],
                           last=[float(c[-1]) if len(c) else None for c in chunks], meta=_plain(meta))
                _close(got, units["chunks"][f"{key}/{md}"], f"chunks[{key}/{md}]")
            if spans:
                m = fvad.SpeechTimestampsMap(spans, 16000)
                want = units["ts_map"][key]
                qs = want["queries"]
                _close([m.get_original_time(q) for q in qs], want["plain"], f"ts_map[{key}].plain")
                _close([m.get_original_time(q, is_end=True) for q in qs], want["ends"], f"ts_map[{key}].ends")
                _close([m.get_chunk_index(q) for q in qs], want["index"], f"ts_map[{key}].index")
            checked += 1
    assert checked == len(hs.vad_prob_tracks()) * len(hs.VAD_CASES)
    assert any(len(v) > 2 for v in units["vad"].values())      # the fixtures are not trivially empty


def test_vad_needs_probabilities(monkeypatch):
    monkeypatch.setenv(fvad.ONNX_ENV, "/nonexistent/silero_vad_v6.onnx")   # no weights on this box
    monkeypatch.setattr(fvad, "_VAD_MODEL", None)
    with pytest.raises(RuntimeError, match="Silero"):
        fvad.get_speech_timestamps(np.zeros(16000, np.float32))
    # a callable model is accepted and sees the padded audio
    seen = {}

    def model(padded):
        seen["n"] = len(padded)
        return np.full(len(padded) // 512, 0.9)
    spans = fvad.get_speech_timestamps(np.zeros(1024, np.float32), vad_model=model)
    assert seen["n"] == 1536 and spans == [{"start": 0, "end": 1024}]


def test_word_splitting_and_punctuation_merge(gold, hf_tok):
    _, units = gold
    cfg = get_config("micro")
    n = 0
    for lang, texts in hs.SPLIT_TEXTS.items():
        tok = Tokenizer(hf_tok, cfg, True, task="transcribe", language=lang)
        for text in texts:
            want = units["split"][f"{lang}|{text}"]
            ids = tok.encode(text) + [tok.timestamp_begin + 10] + tok.encode(" and") + [tok.eot]
            assert ids == want["ids"]
            assert tok.decode_with_timestamps(ids) == want["decoded"]
            words, groups = tok.split_to_word_tokens(ids)
            assert words == want["words"] and groups == want["groups"], (lang, text, words, want["words"])
            alignment = [dict(word=w, tokens=list(g)) for w, g in zip(words, groups)]
            fwords.merge_punctuations(alignment, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、")
            assert alignment == units["merge"][f"{lang}|{text}"]
            n += 1
    assert n == sum(len(t) for t in hs.SPLIT_TEXTS.values())


@pytest.mark.parametrize("name", ["bat_clips_words", "bat_vad"])
def test_batches_in_flight_keep_the_serial_result(gold, hf_tok, name):
    """worker replicas (backend.inter_threads > 1): several batches are decoded concurrently, the segments —
    including the word-timestamp heuristics that chain from batch to batch — stay those of the serial run"""
    scen, _ = gold
    sc = hs.SCENARIOS[name]
    model = make_model(get_config("micro"), hf_tok)
    model.model.inter_threads = 3
    audio = hs.synth_audio(*sc["audio"])
    kwargs = json.loads(json.dumps(sc["kwargs"]))
    kwargs["batch_size"] = 1                                    # many batches -> real overlap
    if kwargs.get("vad_filter"):
        kwargs["vad_speech_probs"] = hs.speech_probs(np.pad(audio, (0, 512 - audio.shape[0] % 512)))
    segments, _ = BatchedInferencePipeline(model).transcribe(audio, **kwargs)
    segments = [_plain(s) for s in segments]
    want = scen[name]["segments"]
    assert len(segments) == len(want)
    for i, (g, w) in enumerate(zip(segments, want)):
        _close(g, w, f"{name}.segments[{i}]")
    assert [c[0] for c in model.model.calls].count("generate") >= 4


def test_empty_audio(hf_tok):
    """the reference's tests/test_transcribe.py:91-97 on the host code: an empty recording yields no segments on either
    driver, and language detection still answers"""
    model = make_model(get_config("micro"), hf_tok)
    audio = np.asarray([], dtype="float32")
    assert list(model.transcribe(audio)[0]) == []
    segs, info = BatchedInferencePipeline(model).transcribe(audio, vad_speech_probs=np.zeros(0, dtype=np.float32))
    assert list(segs) == [] and info.duration == 0.0
    lang, prob, all_probs = model.detect_language(audio)
    assert lang in ("zh", "es", "en", "de") and 0.0 < prob <= 1.0 and len(all_probs) >= 1
