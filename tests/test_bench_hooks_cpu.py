"""CPU tests of two pieces of bench.py that otherwise only run on the GPU box: the `--vad` form of the `pipeline`
measurement (the recording with digital silences, the calibrated synthetic Silero network — here on the host path —, VAD
chunking inside the call) on the scripted backend of tests/test_bench_dist_gloo.py, and the all-cores `cpu_baseline`
(several concurrent oracle streams) on the micro geometry."""
import numpy as np

import bench
import test_bench_dist_gloo as seam_mod


def _fake_backend(cfg, workers=2):
    ns = {}
    exec(seam_mod.SEAM, ns)
    return ns["FakeBackend"](cfg, workers, None)


def test_pipeline_with_vad_on_scripted_backend(monkeypatch):
    from faster_whisper_amd import get_config
    from faster_whisper_amd import vad as fvad
    monkeypatch.setenv("FWAMD_BENCH_VAD_DEVICE", "cpu")
    monkeypatch.setattr(fvad, "_VAD_MODEL", None)
    cfg = get_config("micro")
    n = 6
    out = bench.pipeline_rtf(_fake_backend(cfg), cfg, n, 4, 5, 12, vad=True)
    assert "error" not in out, out
    # every 30 s slot is one burst of 27.5 s + 2.5 s of digital silence: one span and one chunk per slot
    assert out["segments"] == n and out["tokens"] == 12 * n and "native device VAD" in out["vad"]
    assert 27.4 * n < out["duration_after_vad_s"] < 28.6 * n, out
    # the calibrated network separates the calibration clip's silence from its noise
    m = bench.synthetic_vad(device="cpu")
    clip = np.concatenate([np.zeros(512 * 40, np.float32), bench.synth_chunks(1, seed=9)[0][:512 * 40]])
    p = m(clip)
    # (the first windows carry the LSTM's start-up transient, the windows after the edge its memory of the silence)
    assert p[10:38].max() < 0.2 and np.median(p[46:]) > 0.9, (p[10:38].max(), np.median(p[46:]))


def test_cpu_baseline_all_cores_form(monkeypatch):
    from faster_whisper_amd import get_config, synthetic_weights
    monkeypatch.setenv("FWAMD_CPU_BASELINE_TEAM", "2")          # 4 streams of 2 threads on this 8-thread host
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=1234)
    chunks = bench.synth_chunks(4, seed=1000)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    L = 12
    kw = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup, min_new_tokens=L)
    out = bench.cpu_baseline(cfg, w, chunks, prompt, 5, L, kw)
    import os
    streams = (os.cpu_count() or 1) // 2
    assert out["kind"] == "port" and out["one_stream"]["cores"] == 2 and out["one_stream"]["value"] > 0
    if streams > 1:
        assert out["cores"] == 2 * streams and out["all_cores"]["streams"] == streams
        assert len(out["all_cores"]["per_stream_value"]) == streams
        assert abs(out["value"] - sum(out["all_cores"]["per_stream_value"])) < 1e-2 * out["value"]   # (each rounded to 4 digits)
    else:
        assert out["cores"] == 2 and out["value"] == out["one_stream"]["value"]
