"""Host-side integer logic of the batched path (no GPU): prompt construction, suppress set,
timestamp splitting, pad_or_trim, chunk partitioning, result records.  Expected values are
worked out by hand from the reference's rules (transcribe.py:1024-1101, :1532-1565, :1884-1907)."""
import numpy as np
import pytest

from faster_whisper_amd import get_config
from faster_whisper_amd.sharding import decode_records, encode_records, partition
from faster_whisper_amd.transcribe import (Tokenizer, WhisperModel, get_compression_ratio, get_suppressed_tokens,
                                           pad_or_trim)


def _bare_model(cfg):
    m = WhisperModel.__new__(WhisperModel)
    m.max_length = 448
    m.time_precision = 0.02
    m.input_stride = 2
    m.frames_per_second = 100
    m.tokens_per_second = 50
    return m


def test_tokenizer_special_ids_and_sot_sequence():
    v3 = get_config("large-v3")
    tok = Tokenizer(None, v3, True, task="transcribe", language="fr")
    assert tok.sot_sequence == [50258, 50259 + 6, 50360]   # fr is the 7th language of the vocabulary
    assert tok.timestamp_begin == 50365 and tok.no_timestamps == 50364
    en = get_config("tiny.en")
    tok_en = Tokenizer(None, en, False)
    assert tok_en.sot_sequence == [50257]
    with pytest.raises(ValueError):
        Tokenizer(None, v3, True, task="nope", language="en")
    with pytest.raises(ValueError):
        Tokenizer(None, v3, True, task="transcribe", language="xx")


def test_get_prompt():
    cfg = get_config("large-v3")
    m = _bare_model(cfg)
    tok = Tokenizer(None, cfg, True, task="transcribe", language="en")
    assert m.get_prompt(tok, [], without_timestamps=True) == [50258, 50259, 50360, 50364]
    assert m.get_prompt(tok, [], without_timestamps=False) == [50258, 50259, 50360]
    prev = list(range(1000, 1300))
    p = m.get_prompt(tok, prev, without_timestamps=True)
    assert p[0] == cfg.sot_prev and p[1:224] == prev[-223:] and p[224:] == [50258, 50259, 50360, 50364]


def test_suppressed_tokens_matches_reference_rule():
    en = get_config("tiny.en")
    tok = Tokenizer(None, en, False)
    got = get_suppressed_tokens(tok, [-1])
    # without a vocabulary the non-speech set is empty; the six specials are pinned by the
    # reference's tests/test_tokenizer.py:110
    assert got == (50257, 50357, 50358, 50359, 50360, 50361)
    assert get_suppressed_tokens(tok, [5, 3]) == (3, 5, 50257, 50357, 50358, 50359, 50360, 50361)


def test_split_segments_by_timestamps():
    cfg = get_config("large-v3")
    m = _bare_model(cfg)
    tok = Tokenizer(None, cfg, True, task="transcribe", language="en")
    tb = cfg.timestamp_begin
    # <0.00> a b <2.00><2.00> c <4.50>   -> two segments, single timestamp ending
    tokens = [tb, 11, 12, tb + 100, tb + 100, 13, tb + 225]
    segs, seek, single = m._split_segments_by_timestamps(tok, tokens, time_offset=30.0, segment_size=3000,
                                                         segment_duration=30.0, seek=0)
    assert single is True and seek == 3000
    assert [s["tokens"] for s in segs] == [[tb, 11, 12, tb + 100], [tb + 100, 13, tb + 225]]
    assert segs[0]["start"] == pytest.approx(30.0) and segs[0]["end"] == pytest.approx(32.0)
    assert segs[1]["start"] == pytest.approx(32.0) and segs[1]["end"] == pytest.approx(34.5)
    # consecutive timestamps but no single-timestamp ending: seek advances to the last pair
    tokens = [tb, 11, tb + 50, tb + 50, 12, 13]
    segs, seek, single = m._split_segments_by_timestamps(tok, tokens, 0.0, 3000, 30.0, 0)
    assert single is False and len(segs) == 1 and seek == 50 * 2
    # no consecutive timestamps: one segment, duration from the last timestamp
    tokens = [11, 12, tb + 200]
    segs, seek, _ = m._split_segments_by_timestamps(tok, tokens, 60.0, 3000, 30.0, 0)
    assert len(segs) == 1 and segs[0]["end"] == pytest.approx(64.0) and seek == 3000
    # text only: whole chunk
    segs, seek, _ = m._split_segments_by_timestamps(tok, [11, 12], 0.0, 2500, 25.0, 0)
    assert segs[0]["start"] == 0.0 and segs[0]["end"] == 25.0 and seek == 2500


def test_pad_or_trim_and_compression_ratio():
    a = np.arange(12, dtype=np.float32).reshape(2, 6)
    assert pad_or_trim(a, 4).shape == (2, 4) and np.array_equal(pad_or_trim(a, 4), a[:, :4])
    p = pad_or_trim(a, 9)
    assert p.shape == (2, 9) and np.array_equal(p[:, :6], a) and float(np.abs(p[:, 6:]).max()) == 0.0
    assert get_compression_ratio("a" * 100) > 5


def test_partition_is_contiguous_and_balanced():
    for n, w in [(120, 8), (10, 4), (3, 8), (0, 2), (17, 1)]:
        parts = partition(n, w)
        assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
    assert partition(120, 8)[3] == (45, 60)


def test_result_records_roundtrip():
    class R:
        def __init__(self, ids, s, n):
            self.sequences_ids, self.scores, self.no_speech_prob = [ids], [s], n
    rs = [R([1, 2, 3], -0.25, 0.5), R([], -1.5, 0.0), R(list(range(40)), 0.0, 1.0)]
    rec = encode_records(rs, 32)
    assert rec.shape == (3, 37) and rec.dtype == np.int32   # 1 + 32 ids + 2 float64
    back = decode_records(rec, 32)
    assert back[0] == ([1, 2, 3], -0.25, 0.5) and back[1] == ([], -1.5, 0.0)
    assert back[2][0] == list(range(32))   # truncated to max_len


def test_bench_pipeline_measure_runs_on_a_scripted_backend():
    """bench.py's secondary end-to-end number: exercised here without a GPU (scripted backend), and a broken
    backend must be reported, never raised"""
    import bench
    from oracle import micro_tokenizer
    from oracle.scripted_backend import ScriptedBackend
    cfg = get_config("micro")
    backend = ScriptedBackend(cfg, micro_tokenizer.build())
    backend.inter_threads = 3
    r = bench.pipeline_rtf(backend, cfg, 7, 2, 5, 20)
    assert r["segments"] == 7 and r["tokens"] == 7 * 20 and r["audio_s"] == 210.0 and r["value"] > 0
    assert "error" in bench.pipeline_rtf(object(), cfg, 2, 2, 5, 20)


def test_idle_decode_group_splits_known_work_evenly():
    """Round 5: the run size an IDLE two-lane decode group leads with (csrc/decoder.hip: idle_lead_chunks, through the
    host-only hook fw_test_idle_lead_chunks — no device involved).  The reference's replica pool decodes batches side by
    side (transcribe.py:645-657); here the batches of the workers share decode runs, and a group with no run in progress
    starts its first run at its even share of the work it knows of instead of waiting for 90 % of a run's capacity."""
    from faster_whisper_amd import _lib
    f = _lib.load().fw_test_idle_lead_chunks
    cap = 320                                       # chunks one run takes (1 600 rows at beam 5)
    # the driver's burst: 20 batches of 16 in flight — k queued, 20 - k still encoding -> lead at 10 batches
    for k in range(1, 21):
        assert f(16 * k, k, 20 - k, cap, 16) == 160
    # all 32 workers in flight: 512 chunks > one run -> two runs of 256 (the plain rule would wait for 288)
    assert f(16, 1, 31, cap, 16) == 256
    # more than two runs' worth: ceil(1 000 / 320) = 4 runs of 250
    assert f(200, 10, 40, cap, 20) == 250
    # a lone caller / nothing else on its way: its own batch, at least one max_batch
    assert f(16, 1, 0, cap, 16) == 16
    assert f(3, 1, 0, cap, 16) == 16
    # ragged batches: the pending requests are counted at the average size of the queued ones
    assert f(11 + 16, 2, 2, cap, 16) == (27 + 2 * 13 + 1) // 2
    # degenerate arguments do not divide by zero
    assert f(0, 0, 0, 0, 1) == 1 and f(16, 0, -3, cap, 16) == 16


def test_speech_timestamps_skip_ahead_equals_every_window():
    """round 6: get_speech_timestamps jumps over the windows that cannot change its state (an 8 h recording is 900 000
    windows); the plain walk over every window — the reference's loop, vad.py:85-160 — must give the same spans for any
    probabilities and options: run lengths around every duration threshold, probabilities inside the hysteresis band,
    over-long speech with and without a cut candidate"""
    from faster_whisper_amd.vad import VadOptions, get_speech_timestamps
    rng = np.random.default_rng(2024)
    n_cases = 0
    for case in range(400):
        n = int(rng.integers(1, 1500))
        kind = case % 4
        if kind == 0:                                   # i.i.d. noise around the thresholds
            p = rng.uniform(0.2, 0.8, n)
        elif kind == 1:                                 # long runs of speech / silence / in-band values
            p = np.empty(n)
            i = 0
            while i < n:
                run = int(rng.integers(1, 120))
                p[i:i + run] = rng.choice([0.02, 0.3, 0.42, 0.6, 0.97]) + rng.uniform(-0.01, 0.01)
                i += run
        elif kind == 2:                                 # mostly speech with short dips: exercises the max-duration cuts
            p = np.full(n, 0.9)
            for _ in range(int(rng.integers(0, 12))):
                a = int(rng.integers(0, n))
                p[a:a + int(rng.integers(1, 9))] = rng.choice([0.1, 0.4])
        else:                                           # mostly silence with bursts
            p = np.full(n, 0.05)
            for _ in range(int(rng.integers(0, 10))):
                a = int(rng.integers(0, n))
                p[a:a + int(rng.integers(1, 200))] = 0.8
        opts = VadOptions(threshold=float(rng.choice([0.5, 0.35, 0.6])),
                          neg_threshold=None if rng.random() < 0.6 else float(rng.choice([0.2, 0.3])),
                          min_speech_duration_ms=int(rng.choice([0, 100, 250])),
                          max_speech_duration_s=float(rng.choice([float("inf"), 30.0, 4.0, 1.5])),
                          min_silence_duration_ms=int(rng.choice([160, 2000, 500, 64])),
                          speech_pad_ms=int(rng.choice([400, 30, 0])))
        audio = np.zeros(512 * n - int(rng.integers(0, 512)) if n > 1 else 300, np.float32)
        probs = p[:len(audio) // 512 + 1] if len(audio) // 512 + 1 <= n else p
        fast = get_speech_timestamps(audio, opts, speech_probs=probs)
        slow = get_speech_timestamps(audio, opts, speech_probs=probs, _every_window=True)
        assert fast == slow, (case, opts, fast[:3], slow[:3])
        n_cases += bool(slow)
    assert n_cases > 200          # most cases do produce spans
