"""The Silero VAD v6 network behind the C ABI (host C++, csrc/vad_host.cpp) against
the numpy restatement of the ONNX graph (oracle/silero.py), and the minimal ONNX reader.  No GPU.

Parity with the reference's onnxruntime run is UNPINNED (onnxruntime is absent); with the real asset available
(build container: /root/reference, or $FWAMD_SILERO_VAD_ONNX) the probabilities on the reference's own speech
fixture must be high on speech and low on digital silence and match the committed values."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from faster_whisper_amd import _lib, onnx_lite, vad
from oracle import silero

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_ONNX = "/root/reference/faster_whisper/assets/silero_vad_v6.onnx"


def synthetic_weights(seed=0):
    """random weights with the exact initializer names / shapes of silero_vad_v6.onnx"""
    rng = np.random.default_rng(seed)
    f = lambda *s, scale=0.08: (rng.standard_normal(s) * scale).astype(np.float32)   # noqa: E731
    w = {"encoder.feature_extractor.forward_basis_buffer": f(258, 1, 256, scale=0.05),
         "decoder.conv1d.weight": f(1, 128, 1, scale=0.3), "decoder.conv1d.bias": f(1, scale=0.1),
         "onnx::LSTM_w": f(1, 512, 128), "onnx::LSTM_r": f(1, 512, 128), "onnx::LSTM_b": f(1, 1024, scale=0.2)}
    for i, (co, ci) in enumerate([(128, 129), (64, 128), (64, 64), (128, 64)]):
        w[f"encoder.conv_layers.{i}.weight"] = f(co, ci, 3)
        w[f"encoder.conv_layers.{i}.bias"] = f(co, scale=0.05)
    return w


def test_network_matches_numpy_restatement():
    w = synthetic_weights(3)
    model = vad.SileroVADModel(weights=w, n_threads=3)
    rng = np.random.default_rng(5)
    audio = (rng.standard_normal(512 * 300) * 0.2).astype(np.float32)
    audio[512 * 100:512 * 140] = 0.0
    got = model(audio)
    ref, _, _ = silero.forward(w, silero.frame_windows(audio))
    assert got.shape == (300,) and got.dtype == np.float32
    assert np.all((got > 0) & (got < 1)) and got.std() > 1e-3          # not a degenerate constant
    assert np.abs(got - ref).max() < 2e-5
    # thread count does not change a bit (every window is computed independently, the recurrence is sequential)
    assert np.array_equal(got, vad.SileroVADModel(weights=w, n_threads=1)(audio))


def test_state_is_carried_across_calls():
    """the windows are the LSTM's sequence: two calls with h / c handed over == one call (the reference carries the
    state across its batches of 10 000 windows, vad.py:338-347)"""
    w = synthetic_weights(4)
    model = vad.SileroVADModel(weights=w)
    lib = _lib.load()
    rng = np.random.default_rng(6)
    win = np.ascontiguousarray((rng.standard_normal((50, 576)) * 0.3).astype(np.float32))

    def run(x, h, c):
        p = np.empty(len(x), np.float32)
        _lib.check(lib.fw_vad_forward(model._handle, _lib.ptr(np.ascontiguousarray(x)), len(x), 1, _lib.ptr(h),
                                      _lib.ptr(c), _lib.ptr(p)))
        return p
    h, c = np.zeros(128, np.float32), np.zeros(128, np.float32)
    whole = run(win, h, c)
    h2, c2 = np.zeros(128, np.float32), np.zeros(128, np.float32)
    parts = np.concatenate([run(win[:17], h2, c2), run(win[17:], h2, c2)])
    assert np.array_equal(whole, parts) and np.array_equal(h, h2) and np.array_equal(c, c2)
    ref, rh, rc = silero.forward(w, win)
    assert np.abs(whole - ref).max() < 2e-5 and np.abs(h - rh).max() < 2e-5 and np.abs(c - rc).max() < 2e-5
    # n = 0 is a no-op, bad arguments are reported
    assert lib.fw_vad_forward(model._handle, None, 0, 1, _lib.ptr(h), _lib.ptr(c), None) == 0
    assert lib.fw_vad_forward(model._handle, None, 5, 1, _lib.ptr(h), _lib.ptr(c), None) == -1
    assert b"null" in lib.fw_last_error()


def test_framing_quirk_of_the_reference():
    """context = tail of the previous window (zeros first); the reference's in-place `context[-1] = 0` also
    clears the last 64 samples of the last window"""
    a = np.arange(1, 512 * 3 + 1, dtype=np.float32)
    w = silero.frame_windows(a)
    assert w.shape == (3, 576)
    assert np.all(w[0, :64] == 0) and np.array_equal(w[1, :64], a[512 - 64:512]) and np.array_equal(w[1, 64:], a[512:1024])
    assert np.all(w[2, -64:] == 0) and np.array_equal(w[2, 64:-64], a[1024:1536 - 64])
    assert a[-1] == 512 * 3                                             # the caller's array is not modified


def test_rejects_other_models():
    w = synthetic_weights(1)
    w["encoder.conv_layers.0.weight"] = w["encoder.conv_layers.0.weight"][:, :100]
    with pytest.raises(ValueError, match="Silero VAD v6"):
        vad.SileroVADModel(weights=w)


# ---- minimal protobuf writer, only to test the reader ------------------------------------------------------
def _vi(n):
    out = b""
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out += bytes([b | (0x80 if n else 0)])
        if not n:
            return out


def _ld(field, payload):
    return _vi(field << 3 | 2) + _vi(len(payload)) + payload


def _tensor(name, arr, dtype_id):
    body = b"".join(_vi(1 << 3) + _vi(d) for d in arr.shape)
    return body + _vi(2 << 3) + _vi(dtype_id) + _ld(8, name.encode()) + _ld(9, arr.tobytes())


def test_onnx_reader(tmp_path):
    w = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    pads = np.array([0, 128, 0, -128], dtype=np.int64)
    attr_ints = _ld(1, b"strides") + _ld(8, _vi(128) + _vi(2))                    # packed ints
    attr_f = _ld(1, b"alpha") + _vi(2 << 3 | 5) + struct.pack("<f", 0.25)
    attr_i = _ld(1, b"hidden_size") + _vi(3 << 3) + _vi(128)
    attr_s = _ld(1, b"mode") + _ld(4, b"reflect")
    node = (_ld(1, b"x") + _ld(1, b"w") + _ld(2, b"y") + _ld(4, b"Conv") + _ld(5, attr_ints) + _ld(5, attr_f)
            + _ld(5, attr_i) + _ld(5, attr_s))
    graph = (_ld(1, node) + _ld(5, _tensor("w", w, 1)) + _ld(5, _tensor("pads", pads, 7))
             + _ld(11, _ld(1, b"x")) + _ld(11, _ld(1, b"w")) + _ld(12, _ld(1, b"y")))
    path = tmp_path / "m.onnx"
    path.write_bytes(_vi(1 << 3) + _vi(8) + _ld(7, graph))
    nodes, inits, ins, outs = onnx_lite.load(str(path))
    assert ins == ["x"] and outs == ["y"]
    assert nodes == [dict(op="Conv", inputs=["x", "w"], outputs=["y"],
                          attrs=dict(strides=[128, 2], alpha=0.25, hidden_size=128, mode="reflect"))]
    assert np.array_equal(inits["w"], w) and inits["pads"].tolist() == [0, 128, 0, -128]
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.onnx"
        bad.write_bytes(_vi(1 << 3) + _vi(8))
        onnx_lite.load(str(bad))


def test_missing_weights_fail_loudly(monkeypatch):
    monkeypatch.setenv(vad.ONNX_ENV, "/nonexistent/silero_vad_v6.onnx")
    monkeypatch.setattr(vad, "_VAD_MODEL", None)
    with pytest.raises(RuntimeError, match="Silero VAD weights not found"):
        vad.get_speech_timestamps(np.zeros(16000, np.float32))


def _real_onnx():
    p = os.environ.get(vad.ONNX_ENV) or REF_ONNX
    return p if os.path.isfile(p) else None


@pytest.mark.skipif(_real_onnx() is None, reason="silero_vad_v6.onnx not available on this box")
def test_real_weights_on_the_reference_speech_fixture():
    nodes, inits, ins, outs = onnx_lite.load(_real_onnx())
    assert [n["op"] for n in nodes].count("Conv") == 6 and [n["op"] for n in nodes].count("LSTM") == 1
    assert ins == ["input", "h", "c"] and outs == ["speech_probs", "hn", "cn"]
    speech = np.load(os.path.join(GOLD, "speech_pcm.npz"))["pcm"].astype(np.float32)
    audio = np.concatenate([np.zeros(32000, np.float32), speech, np.zeros(31000, np.float32)])
    padded = np.pad(audio, (0, 512 - len(audio) % 512))
    model = vad.SileroVADModel(_real_onnx())
    probs = model(padded)
    ref, _, _ = silero.forward(inits, silero.frame_windows(padded))
    assert np.abs(probs - ref).max() < 1e-5
    lead, tail = probs[:55], probs[-55:]
    body = probs[70:70 + len(speech) // 512 - 16]
    assert lead.max() < 0.05 and tail[5:].max() < 0.05 and (body > 0.5).mean() > 0.7
    gold = np.load(os.path.join(GOLD, "vad_speech_probs.npy"))
    assert np.abs(probs - gold).max() < 1e-5
    # end to end: the state machine finds one speech span inside the padded recording
    spans = vad.get_speech_timestamps(audio, vad.VadOptions(min_silence_duration_ms=300), vad_model=model)
    assert len(spans) >= 1 and spans[0]["start"] > 16000 and spans[-1]["end"] < len(audio) - 16000


@pytest.mark.skipif(_real_onnx() is None, reason="silero_vad_v6.onnx not available on this box")
def test_detect_language_with_vad_filter(monkeypatch):
    """WhisperModel.detect_language(audio, vad_filter=True): the speech is cut out before the first segments are
    looked at (transcribe.py:1802-1806) — here with the native VAD and a scripted backend"""
    from faster_whisper_amd import get_config
    from oracle import micro_tokenizer
    from test_host_golden import make_model
    monkeypatch.setenv(vad.ONNX_ENV, _real_onnx())
    monkeypatch.setattr(vad, "_VAD_MODEL", None)
    model = make_model(get_config("micro"), micro_tokenizer.build())
    speech = np.load(os.path.join(GOLD, "speech_pcm.npz"))["pcm"].astype(np.float32)
    audio = np.concatenate([np.zeros(48000, np.float32), speech, np.zeros(48000, np.float32)])
    lang, prob, all_probs = model.detect_language(audio=audio, vad_filter=True,
                                                  vad_parameters=dict(min_silence_duration_ms=300))
    lang2, prob2, _ = model.detect_language(audio=audio)
    assert lang in ("en", "zh", "de", "es") and 0 < prob <= 1 and len(all_probs) == 4
    # the silence was removed: the encoder saw a different (shorter, louder) window than without the filter
    enc_calls = [c[1] for c in model.model.calls if c[0] == "encode"]
    assert len(enc_calls) == 2 and enc_calls[0] != enc_calls[1]
