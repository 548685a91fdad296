"""BatchedInferencePipeline.transcribe() end to end on the GPU (reference contract:
faster_whisper/transcribe.py:254-617) against the oracle driven through the same host logic."""
import numpy as np
import pytest

from conftest import bench_audio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wm(tmp_path_factory):
    from faster_whisper_amd import get_config, save_model_dir, synthetic_weights
    from faster_whisper_amd.transcribe import WhisperModel
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=21)
    d = str(tmp_path_factory.mktemp("model") / "micro")
    save_model_dir(d, cfg, w)                       # exercises the model-directory loader
    model = WhisperModel(d, device="cuda", compute_type="float16", max_batch_size=4, max_beam_size=5)
    return cfg, w, model


def _audio(n_chunks=5):
    parts = [bench_audio(480000, seed=10 + i) for i in range(n_chunks)]
    parts[-1] = parts[-1][:300000]                  # ragged last clip (18.75 s)
    audio = np.concatenate(parts)
    clips, t = [], 0.0
    for p in parts:
        clips.append({"start": t, "end": t + len(p) / 16000.0})
        t += len(p) / 16000.0
    return audio, clips


def test_transcribe_matches_oracle(wm):
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    from oracle import logmel as olm
    from oracle.whisper import OracleWhisper
    cfg, w, model = wm
    audio, clips = _audio()
    pipe = BatchedInferencePipeline(model)
    kw = dict(language="en", beam_size=5, batch_size=2, clip_timestamps=clips, max_new_tokens=12,
              without_timestamps=False, suppress_tokens=[1, 2, 3])
    segs, info = pipe.transcribe(audio, **kw)
    segs = list(segs)
    assert info.language == "en" and info.duration == pytest.approx(len(audio) / 16000.0)
    assert [s.id for s in segs] == list(range(1, len(segs) + 1))
    # oracle through the same host arithmetic
    oracle = OracleWhisper(cfg, w, emulate_fp16=True)
    tok = model.make_tokenizer(task="transcribe", language="en")
    prompt = model.get_prompt(tok, [], without_timestamps=False)
    chunks = [audio[int(c["start"] * 16000):int(c["end"] * 16000)] for c in clips]
    feats = olm.log_mel_chunks(chunks, cfg.n_mels)
    enc = oracle.encode(feats)
    sup = info.transcription_options.suppress_tokens
    ref = oracle.generate(enc, [prompt] * len(chunks), beam_size=5, max_length=len(prompt) + 12, suppress_tokens=sup,
                          max_initial_timestamp_index=50)
    by_chunk = {}
    for s in segs:
        by_chunk.setdefault(s.seek, []).append(s)
    agree = 0
    for c, r in zip(clips, ref):
        got = by_chunk[int(c["start"] * 100)]
        toks = [t for s in got for t in s.tokens]
        n = len(r.sequences_ids[0])
        exp_avg = r.scores[0] * n / (n + 1)          # transcribe.py:241-246
        if toks == r.sequences_ids[0]:
            agree += 1
            assert got[0].avg_logprob == pytest.approx(exp_avg, abs=2e-3)
        assert got[0].start >= c["start"] - 1e-6 and got[-1].end <= c["end"] + 30.0
        assert all(0.0 <= s.no_speech_prob <= 1.0 for s in got)
    print(f"pipeline: {agree}/{len(clips)} chunks token-identical to the oracle")
    assert agree >= len(clips) - 1


def test_fused_and_host_feature_paths_agree(wm):
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    cfg, w, model = wm
    audio, clips = _audio(3)
    pipe = BatchedInferencePipeline(model)
    kw = dict(language="en", beam_size=1, batch_size=3, clip_timestamps=clips, max_new_tokens=8)
    a = list(pipe.transcribe(audio, fused_features=True, **kw)[0])
    b = list(pipe.transcribe(audio, fused_features=False, **kw)[0])
    assert [s.tokens for s in a] == [s.tokens for s in b]
    assert [s.avg_logprob for s in a] == pytest.approx([s.avg_logprob for s in b], abs=1e-5)


def test_language_detection_and_errors(wm, monkeypatch):
    from faster_whisper_amd import vad as fvad
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    monkeypatch.setenv(fvad.ONNX_ENV, "/nonexistent/silero_vad_v6.onnx")   # the Silero weights are not on this box
    monkeypatch.setattr(fvad, "_VAD_MODEL", None)
    cfg, w, model = wm
    audio, clips = _audio(2)
    pipe = BatchedInferencePipeline(model)
    segs, info = pipe.transcribe(audio, beam_size=1, batch_size=2, clip_timestamps=clips, max_new_tokens=4)
    assert info.language in model.supported_languages and 0.0 < info.language_probability <= 1.0
    assert len(info.all_language_probs) == cfg.n_langs
    list(segs)
    with pytest.raises(ValueError):                     # prompt + max_new_tokens > 448 (transcribe.py:198-207)
        list(pipe.transcribe(audio, language="en", clip_timestamps=clips, max_new_tokens=447)[0])
    with pytest.raises(RuntimeError):                   # > 30 s without clips and without VAD
        pipe.transcribe(audio, language="en", vad_filter=False)
    short = audio[:160000]
    segs, _ = pipe.transcribe(short, language="en", beam_size=1, max_new_tokens=4, vad_filter=False)  # < 30 s: one clip
    assert len(list(segs)) >= 1
    with pytest.raises(RuntimeError, match="Silero"):   # default vad_filter=True needs the VAD weights (or probabilities)
        pipe.transcribe(short, language="en", beam_size=1, max_new_tokens=4)
    # ... which can be supplied: all-speech probabilities keep the whole clip
    probs = np.full(len(short) // 512 + 1, 0.9, dtype=np.float32)
    segs, info = pipe.transcribe(short, language="en", beam_size=1, max_new_tokens=4, vad_speech_probs=probs)
    assert len(list(segs)) >= 1 and info.duration_after_vad == pytest.approx(10.0)


def test_worker_replicas_share_weights(wm):
    """inter_threads > 1: two replicas on one GPU, concurrent generate() from two threads, same answers"""
    import threading
    from faster_whisper_amd import Whisper
    cfg, w, _ = wm
    model = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": w}, inter_threads=2,
                    max_batch_size=2, max_beam_size=5)
    chunks = [bench_audio(480000, seed=3), bench_audio(480000, seed=4)]
    prompt = [cfg.sot, cfg.lang_begin, cfg.transcribe, cfg.no_timestamps]
    out = {}

    def work(i):
        enc = model.encode_pcm(chunks)
        out[i] = [r.sequences_ids[0] for r in model.generate(enc, [prompt] * 2, beam_size=5, max_length=20)]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert out[0] == out[1] and len(out[0][0]) == 16


def test_batches_larger_than_the_engine_workspace(wm):
    """CTranslate2 accepts any batch size; the engine's workspaces hold max_batch_size chunks, so the front splits a
    larger batch into sub-batches (encode, generate, detect_language, align) and concatenates the results"""
    from faster_whisper_amd import Whisper
    cfg, w, _ = wm
    model = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": w}, max_batch_size=2,
                    max_beam_size=5)
    chunks = [bench_audio(480000 if i % 2 else 300000, seed=20 + i) for i in range(5)]
    prompt = [cfg.sot, cfg.lang_begin, cfg.transcribe, cfg.no_timestamps]
    enc = model.encode_pcm(chunks)
    assert enc.shape == [5, 1500, cfg.d_model] and enc.to_numpy().shape == (5, 1500, cfg.d_model)
    kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True, return_no_speech_prob=True)
    got = model.generate(enc, [prompt] * 5, **kw)
    lang = model.detect_language(enc)
    al = model.align(enc, cfg.sot_sequence, [[11, 12, 13]] * 5, [3000, 1800, 3000, 1800, 3000])
    assert len(got) == len(lang) == len(al) == 5
    for i in range(5):
        e1 = model.encode_pcm(chunks[i:i + 1])
        one = model.generate(e1, [prompt], **kw)[0]
        assert got[i].sequences_ids == one.sequences_ids and got[i].scores == one.scores
        assert lang[i] == model.detect_language(e1)[0]
    feats = model.log_mel(chunks)
    assert model.encode(feats).shape == [5, 1500, cfg.d_model]
    with pytest.raises(ValueError):
        model.generate(enc, [prompt] * 5, beam_size=6)          # beam_size is bounded by max_beam_size


def test_empty_audio(wm):
    """the reference's tests/test_transcribe.py:91-97: an empty recording yields no segments on either driver and
    language detection still answers — here on the engine (zero-sample log-mel, one all-padding window)"""
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    cfg, w, model = wm
    audio = np.asarray([], dtype="float32")
    assert list(model.transcribe(audio)[0]) == []
    no_probs = np.zeros(0, dtype=np.float32)                # the Silero pass over zero windows
    segs, info = BatchedInferencePipeline(model).transcribe(audio, vad_speech_probs=no_probs)
    assert list(segs) == [] and info.duration == 0.0
    lang, prob, all_probs = model.detect_language(audio)
    assert lang in model.supported_languages and 0.0 < prob <= 1.0 and len(all_probs) == cfg.n_langs


def test_transcribe_reads_flac_files(wm):
    """the reference accepts a path (transcribe.py:278, decode_audio): a FLAC file goes through the native decoder
    (csrc/flac_host.cpp) and must give exactly what the decoded waveform gives when it is passed as an array"""
    import os
    from conftest import GOLDEN
    from faster_whisper_amd.audio import decode_audio
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    cfg, w, model = wm
    path = os.path.join(GOLDEN, "flac_jfk_head.flac")      # 0.84 s of the reference's jfk.flac (44.1 kHz stereo 24 bit)
    wave = decode_audio(path)
    assert wave.dtype == np.float32 and wave.shape == (13375,)
    pipe = BatchedInferencePipeline(model)
    kw = dict(language="en", beam_size=2, batch_size=2, clip_timestamps=[{"start": 0.0, "end": len(wave) / 16000.0}],
              max_new_tokens=10)
    a, ia = pipe.transcribe(path, **kw)
    b, ib = pipe.transcribe(wave, **kw)
    a, b = list(a), list(b)
    assert len(a) == len(b) >= 1 and ia.duration == ib.duration == pytest.approx(13375 / 16000.0)
    for x, y in zip(a, b):
        assert x.tokens == y.tokens and x.avg_logprob == y.avg_logprob and (x.start, x.end) == (y.start, y.end)
    with open(path, "rb") as f:                            # file objects too
        c, _ = model.transcribe(f, language="en", beam_size=1, max_new_tokens=6)
        assert len(list(c)) >= 1
