"""Silero VAD on the GPU (csrc/vad.hip: one workgroup per window for the front end, one persistent workgroup for the
LSTM recurrence) against the host C++ path (csrc/vad_host.cpp), which is itself pinned to the numpy restatement of
the reference's ONNX graph (tests/test_vad_network.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_vad_matches_host():
    from faster_whisper_amd import vad
    from test_vad_network import synthetic_weights
    w = synthetic_weights(7)
    host = vad.SileroVADModel(weights=w)
    dev = vad.SileroVADModel(weights=w, device="cuda")
    rng = np.random.default_rng(1)
    audio = (rng.standard_normal(512 * 1000) * 0.2).astype(np.float32)
    audio[512 * 300:512 * 420] = 0.0
    a, b = host(audio), dev(audio)
    assert a.shape == b.shape == (1000,)
    assert np.abs(a - b).max() < 5e-5
    assert np.array_equal(dev(audio), b)          # deterministic
