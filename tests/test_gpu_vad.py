"""Silero VAD on the GPU (csrc/vad.hip: one workgroup per window for the front end, one persistent workgroup for the
LSTM recurrence) against the ORACLE — the numpy restatement of the reference's ONNX graph (oracle/silero.py,
faster_whisper/vad.py:295-351) — directly, through the C ABI (fw_vad_forward_dev), and against the host C++ path
(csrc/vad_host.cpp), which tests/test_vad_network.py pins to the same oracle on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _audio(n_windows, seed=1):
    rng = np.random.default_rng(seed)
    audio = (rng.standard_normal(512 * n_windows) * 0.2).astype(np.float32)
    audio[512 * (n_windows * 3 // 10):512 * (n_windows * 42 // 100)] = 0.0      # a stretch of digital silence
    t = np.arange(512 * (n_windows // 10)) / 16000.0
    audio[:t.shape[0]] += (0.3 * np.sin(2 * np.pi * 180.0 * t)).astype(np.float32)
    return audio


def test_device_vad_matches_the_oracle():
    from faster_whisper_amd import vad
    from oracle import silero
    from test_vad_network import synthetic_weights
    w = synthetic_weights(7)
    dev = vad.SileroVADModel(weights=w, device="cuda")
    audio = _audio(1000)
    ref, rh, rc = silero.forward(w, silero.frame_windows(audio))
    got = dev(audio)
    assert got.shape == ref.shape == (1000,)
    err = float(np.abs(got - ref).max())
    print(f"device VAD vs oracle/silero.py over 1000 windows: max abs err {err:.2e}")
    assert err < 1e-4                              # fp32 on both sides: summation order only
    assert np.array_equal(dev(audio), got)         # deterministic


def test_device_vad_carries_the_lstm_state_like_the_oracle():
    """the reference feeds batches of 10 000 windows and carries h / c across them (vad.py:324-347): two device calls
    with the state handed over equal one oracle pass over the whole sequence"""
    import ctypes as C
    from faster_whisper_amd import _lib, vad
    from oracle import silero
    from test_vad_network import synthetic_weights
    w = synthetic_weights(11)
    dev = vad.SileroVADModel(weights=w, device="cuda")
    win = silero.frame_windows(_audio(600, seed=3))
    ref, rh, rc = silero.forward(w, win)
    lib = _lib.load()
    h = np.zeros(128, np.float32)
    c = np.zeros(128, np.float32)
    out = np.empty(600, np.float32)
    for a, b in ((0, 250), (250, 600)):
        part = np.ascontiguousarray(win[a:b])
        _lib.check(lib.fw_vad_forward_dev(dev._handle, 0, _lib.ptr(part), b - a, _lib.ptr(h), _lib.ptr(c),
                                          _lib.ptr(out[a:b])))
    assert np.abs(out - ref).max() < 1e-4
    assert np.abs(h - rh).max() < 1e-4 and np.abs(c - rc).max() < 1e-4


def test_device_vad_matches_host():
    from faster_whisper_amd import vad
    from test_vad_network import synthetic_weights
    w = synthetic_weights(7)
    host = vad.SileroVADModel(weights=w)
    dev = vad.SileroVADModel(weights=w, device="cuda")
    audio = _audio(1000)
    a, b = host(audio), dev(audio)
    assert a.shape == b.shape == (1000,)
    assert np.abs(a - b).max() < 5e-5


def test_device_framing_equals_host_framing():
    """round 6: fw_vad_forward_audio_dev frames the recording on the device (context = tail of the previous window, zeros
    for the first, last 64 samples of the last window zeroed: faster_whisper/vad.py:318-336); the probabilities must be
    the ones the host-framed rows give through fw_vad_forward_dev, bit for bit, and the LSTM state too"""
    import ctypes as C
    from faster_whisper_amd import _lib, vad
    from oracle import silero
    from test_vad_network import synthetic_weights
    w = synthetic_weights(5)
    dev = vad.SileroVADModel(weights=w, device="cuda")
    lib = _lib.load()
    for n_win in (1, 2, 37, 800):
        audio = _audio(max(n_win, 10), seed=n_win)[:512 * n_win]
        win = silero.frame_windows(audio)
        assert win.shape == (n_win, 576)
        out = []
        for mode in (0, 1):
            h = np.zeros(128, np.float32)
            c = np.zeros(128, np.float32)
            p = np.empty(n_win, np.float32)
            if mode == 0:
                _lib.check(lib.fw_vad_forward_dev(dev._handle, 0, _lib.ptr(np.ascontiguousarray(win)), n_win, _lib.ptr(h),
                                                  _lib.ptr(c), _lib.ptr(p)))
            else:
                a = np.ascontiguousarray(audio)
                _lib.check(lib.fw_vad_forward_audio_dev(dev._handle, 0, _lib.ptr(a), a.shape[0], _lib.ptr(h), _lib.ptr(c),
                                                        _lib.ptr(p)))
            out.append((p, h, c))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert np.array_equal(out[0][2], out[1][2])
        assert np.array_equal(dev(audio), out[1][0])           # what SileroVADModel.__call__ now takes
    with pytest.raises(ValueError):
        a = np.zeros(700, np.float32)
        h = np.zeros(128, np.float32)
        _lib.check(lib.fw_vad_forward_audio_dev(dev._handle, 0, _lib.ptr(a), 700, _lib.ptr(h), _lib.ptr(h), _lib.ptr(a)))
