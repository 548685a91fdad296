import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from faster_whisper_amd import _lib
        return _lib.load().fw_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the driver records
    # silent fallbacks as "native code not loaded".
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


def make_model(name="micro", seed=7, max_batch=4, max_beam=5, compute_type="float16", cfg=None, weights=None):
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    cfg = cfg or get_config(name)
    weights = weights if weights is not None else synthetic_weights(cfg, seed=seed)
    model = Whisper(f"synthetic:{name}", device="cuda", files={"config": cfg, "weights": weights},
                    compute_type=compute_type, max_batch_size=max_batch, max_beam_size=max_beam)
    return cfg, weights, model


def bench_audio(n_samples=480000, seed=0):
    """SURVEY.md section 8d synthetic audio: 0.1*N(0,1) + three partials (220/440/880 Hz, amp 0.05)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / 16000.0
    x = 0.1 * rng.standard_normal(n_samples)
    for f in (220.0, 440.0, 880.0):
        x += 0.05 * np.sin(2 * np.pi * f * t)
    return x.astype(np.float32)
