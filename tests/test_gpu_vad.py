"""Silero VAD on the GPU (csrc/vad.hip) against the host C++ path.  The kernels were written after round 1's GPU
budget was spent and have not run on hardware yet, so this test is opt-in (FWAMD_TEST_UNVALIDATED=1): an
unvalidated kernel must not be able to hang the regular `-m gpu` run.  Remove the gate once it has passed."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("FWAMD_TEST_UNVALIDATED") != "1",
                                 reason="device VAD not yet validated on hardware (set FWAMD_TEST_UNVALIDATED=1)")]


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
