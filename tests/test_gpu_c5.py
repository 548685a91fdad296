"""BASELINE config C5 as ONE composition on the GPU (round 6, verdict item 2): BatchedInferencePipeline.transcribe with
`vad_filter=True` on the NATIVE device VAD (csrc/vad.hip, fw_vad_forward_dev) + `word_timestamps=True` + batch_size=16 —
the reference's path transcribe.py:399-402 (VAD chunking) -> :161-170 (batched word timestamps over ragged chunks,
`num_frames` as a list) -> vad.py:186-243 (collect_chunks / restore_speech_timestamps).  Every piece has its own test
(test_gpu_vad.py, test_gpu_pipeline.py, test_gpu_sequential.py, the CPU host goldens); here they run TOGETHER:

  * against the same host code driven by the CPU oracle (oracle/oracle_backend.py) on the same VAD probabilities:
    tokens, segment times, avg_logprob, words and word boundaries;
  * against the `clip_timestamps` call fed the VAD's own chunks (the silence-free recording with one clip per collected
    chunk) mapped back through restore_speech_timestamps: exactly equal, field for field;
  * at the distil-large-v3 geometry (32 encoder layers, 2 decoder layers: C5's model) against the oracle.

Synthetic weights: the Whisper side as everywhere; the Silero network with random weights whose output layer is scaled
so that digital silence and noise land on opposite sides of the threshold (a random network's probabilities sit within
+-0.02 of a constant) — the recording has real silences, the chunks are ragged and span several speech spans each."""
import logging

import numpy as np
import pytest

from conftest import bench_audio

pytestmark = pytest.mark.gpu


def peaked_vad_weights(seed=3, gain=400.0, mid=0.10):
    """test_vad_network.synthetic_weights with the decoder's 1x1 convolution scaled: logit' = gain * (logit - mid).
    For seed 3 digital silence gives logit 0.0897 and the bench's noise 0.110 ... 0.175 (host network, measured), so
    silence -> p ~ 0.02 and noise -> p > 0.98."""
    from test_vad_network import synthetic_weights
    w = synthetic_weights(seed)
    w["decoder.conv1d.weight"] = (w["decoder.conv1d.weight"] * gain).astype(np.float32)
    w["decoder.conv1d.bias"] = (w["decoder.conv1d.bias"] * gain - gain * mid).astype(np.float32)
    return w


def recording(spans, seed=60):
    """noise bursts (conftest.bench_audio: 0.1 N(0,1) + partials) separated by digital silence; spans = [(speech s, gap s)]"""
    parts = []
    for i, (sp, gap) in enumerate(spans):
        parts.append(bench_audio(int(16000 * sp), seed=seed + i))
        parts.append(np.zeros(int(16000 * gap), np.float32))
    return np.concatenate(parts)


SPANS = [(6.5, 1.2), (11.0, 2.0), (4.25, 0.8), (9.0, 1.5), (13.5, 2.5), (3.0, 0.7), (17.0, 1.1), (8.0, 3.0), (5.5, 0.9),
         (12.0, 0.5)]


def _shell(backend, cfg, tok):
    from faster_whisper_amd.transcribe import FeatureExtractor, WhisperModel
    wm = WhisperModel.__new__(WhisperModel)
    wm.logger = logging.getLogger("c5-host")
    wm.model = backend
    wm.hf_tokenizer = tok
    wm.feature_extractor = FeatureExtractor(feature_size=cfg.n_mels, backend=backend)
    wm.input_stride, wm.time_precision, wm.max_length = 2, 0.02, 448
    wm.num_samples_per_token = 320
    wm.frames_per_second, wm.tokens_per_second = 100, 50
    return wm


@pytest.fixture(scope="module")
def device_vad():
    from faster_whisper_amd import vad as fvad
    w = peaked_vad_weights()
    dev = fvad.SileroVADModel(weights=w, device="cuda")
    return w, dev


@pytest.fixture(scope="module")
def pair():
    from faster_whisper_amd import get_config, synthetic_weights
    from faster_whisper_amd.transcribe import WhisperModel
    from oracle import micro_tokenizer
    from oracle.oracle_backend import OracleBackend
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=33)
    tok = micro_tokenizer.build()
    gpu = WhisperModel("synthetic:micro", device="cuda", compute_type="float16",
                       files={"config": cfg, "weights": w, "tokenizer.json": tok.to_str().encode()},
                       max_batch_size=16, max_beam_size=5)
    cpu = _shell(OracleBackend(cfg, w, emulate_fp16=True), cfg, tok)
    return cfg, gpu, cpu


def _vad_probs(dev, audio):
    return dev(np.pad(audio, (0, 512 - len(audio) % 512)))


def _by_chunk(segs):
    out = {}
    for s in segs:
        out.setdefault(s.seek, []).append(s)
    return [out[k] for k in sorted(out)]


def _compare_with_oracle(got, ref, tag, min_chunks_equal):
    g, r = _by_chunk(got), _by_chunk(ref)
    assert len(g) == len(r), (len(g), len(r))
    same = 0
    worst_word = worst_seg = worst_lp = worst_p = 0.0
    for a, b in zip(g, r):
        if [s.tokens for s in a] != [s.tokens for s in b]:
            continue                      # a near-tied beam (synthetic weights): everything after it differs legitimately
        same += 1
        for x, y in zip(a, b):
            worst_seg = max(worst_seg, abs(x.start - y.start), abs(x.end - y.end))
            worst_lp = max(worst_lp, abs(x.avg_logprob - y.avg_logprob) / max(1.0, abs(y.avg_logprob)))
            assert x.no_speech_prob == pytest.approx(y.no_speech_prob, abs=1e-3)
            assert [w.word for w in x.words] == [w.word for w in y.words]
            for wa, wb in zip(x.words, y.words):
                worst_word = max(worst_word, abs(wa.start - wb.start), abs(wa.end - wb.end))
                worst_p = max(worst_p, abs(wa.probability - wb.probability))
    print(f"{tag}: {same}/{len(r)} VAD chunks token-identical to the oracle-driven host run; on those: segment times "
          f"within {worst_seg:.3f} s, word boundaries within {worst_word:.3f} s, avg_logprob within {worst_lp:.1e} (rel), "
          f"word probabilities within {worst_p:.1e}")
    assert same >= min_chunks_equal, (same, len(r))
    assert worst_word <= 0.021 and worst_seg <= 0.021, (worst_word, worst_seg)      # one encoder frame (20 ms)
    assert worst_lp < 2e-3 and worst_p < 2e-3, (worst_lp, worst_p)


def test_c5_batched_vad_word_timestamps_micro(pair, device_vad, monkeypatch):
    from faster_whisper_amd import vad as fvad
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    from faster_whisper_amd.words import restore_speech_timestamps
    from oracle import silero
    cfg, gpu, cpu = pair
    vw, dev = device_vad
    monkeypatch.setattr(fvad, "_VAD_MODEL", dev)          # what get_vad_model() hands to get_speech_timestamps
    audio = recording(SPANS)
    # (synthetic weights: three quarters of micro's vocabulary are timestamp ids, which a random decoder emits freely even
    #  after <|notimestamps|>; suppressing them leaves text tokens, i.e. words to time)
    kw = dict(language="en", beam_size=5, batch_size=16, word_timestamps=True, max_new_tokens=14,
              suppress_tokens=[1, 2, 3] + list(range(cfg.timestamp_begin, cfg.n_vocab)), vad_filter=True)
    # ---- A: the product call: native device VAD inside, ragged chunks, align over them ----
    segs, info = BatchedInferencePipeline(gpu).transcribe(audio, **kw)
    got = list(segs)
    probs = _vad_probs(dev, audio)
    ref_probs, _, _ = silero.forward(vw, silero.frame_windows(np.pad(audio, (0, 512 - len(audio) % 512))))
    print(f"device VAD vs oracle/silero.py on the recording: max abs diff {np.abs(probs - ref_probs).max():.2e}; "
          f"{(probs > 0.5).mean():.2f} of the windows are speech")
    # the peaked network separates silence from noise: its spans are the recording's bursts (within speech_pad_ms = 400
    # and the 32 ms window grid)
    spans = fvad.get_speech_timestamps(audio, info.vad_options, speech_probs=probs)
    assert len(spans) == len(SPANS), (len(spans), spans)
    t = 0.0
    for sp, (dur, gap) in zip(spans, SPANS):
        assert abs(sp["start"] / 16000 - t) < 0.55 and abs(sp["end"] / 16000 - (t + dur)) < 0.55, (sp, t, dur)
        t += dur + gap
    chunks, meta = fvad.collect_chunks(audio, spans, max_duration=30)
    assert len(chunks) >= 4 and len({len(c) for c in chunks}) > 1                 # ragged, several spans per chunk
    assert info.duration == pytest.approx(len(audio) / 16000.0)
    assert info.duration_after_vad == pytest.approx(sum(s["end"] - s["start"] for s in spans) / 16000.0)
    assert len(_by_chunk(got)) == len(chunks)
    for s in got:
        assert s.words, "word_timestamps=True: every segment carries words"
        for w in s.words:
            # (a word the DTW puts at the very end of the last chunk may end a little past the recording: the host logic —
            #  pinned to the reference's by tests/golden/host_*.json — does not clip restored word times to the duration)
            assert 0.0 <= w.start <= w.end <= info.duration + 1.0 and 0.0 <= w.probability <= 1.0
    # ---- B: the same host code on the CPU oracle, fed the engine's VAD probabilities ----
    ref = list(BatchedInferencePipeline(cpu).transcribe(audio, vad_speech_probs=probs, **kw)[0])
    _compare_with_oracle(got, ref, "[C5 micro]", min_chunks_equal=len(chunks) - 1)
    # ---- C: clip_timestamps fed the VAD's own chunks: the silence-free recording, one clip per collected chunk ----
    free = np.concatenate(chunks)
    clips = [{"start": (round(m["offset"] * 16000) + 0.5) / 16000.0,
              "end": (round((m["offset"] + m["duration"]) * 16000) + 0.5) / 16000.0} for m in meta]
    kc = dict(kw, vad_filter=False, clip_timestamps=clips)
    via_clips = list(restore_speech_timestamps(iter(list(BatchedInferencePipeline(gpu).transcribe(free, **kc)[0])),
                                               spans, 16000))
    assert len(via_clips) == len(got)
    for a, b in zip(got, via_clips):
        assert a.tokens == b.tokens and a.avg_logprob == b.avg_logprob and a.no_speech_prob == b.no_speech_prob
        assert (a.start, a.end) == (b.start, b.end), (a.start, a.end, b.start, b.end)
        assert [(w.word, w.start, w.end, w.probability) for w in a.words] == \
               [(w.word, w.start, w.end, w.probability) for w in b.words]


def test_c5_distil_large_v3_geometry(device_vad, monkeypatch):
    """the same composition at C5's model shape: distil-large-v3 (128 mels, 32 encoder layers, 2 decoder layers, d = 1280),
    synthetic weights, no tokenizer file (ids render as <id>: every token is its own word)"""
    import os
    import torch
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    from faster_whisper_amd import vad as fvad
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    from oracle.oracle_backend import OracleBackend
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_config("distil-large-v3")
    w = synthetic_weights(cfg, seed=1234)
    vw, dev = device_vad
    monkeypatch.setattr(fvad, "_VAD_MODEL", dev)
    backend = Whisper("synthetic:distil-large-v3", device="cuda", files={"config": cfg, "weights": w},
                      max_batch_size=16, max_beam_size=5)
    gpu = _shell(backend, cfg, None)
    audio = recording(SPANS[:6], seed=80)            # ~57 s of bursts: three ragged VAD chunks
    kw = dict(language="en", beam_size=5, batch_size=16, word_timestamps=True, max_new_tokens=10,
              suppress_tokens=[cfg.sot, cfg.no_speech] + list(range(cfg.timestamp_begin, cfg.n_vocab)), vad_filter=True)
    got = list(BatchedInferencePipeline(gpu).transcribe(audio, **kw)[0])
    probs = _vad_probs(dev, audio)
    cpu = _shell(OracleBackend(cfg, w, emulate_fp16=True), cfg, None)
    ref = list(BatchedInferencePipeline(cpu).transcribe(audio, vad_speech_probs=probs, **kw)[0])
    n = len(_by_chunk(ref))
    assert n >= 2
    _compare_with_oracle(got, ref, "[C5 distil-large-v3]", min_chunks_equal=n - 1)
