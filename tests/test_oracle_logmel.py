"""Pins the numpy log-mel oracle against golden vectors produced by the REFERENCE
(faster_whisper/feature_extractor.py, via oracle/gen_golden.py) — SURVEY.md section 8c."""
import numpy as np
import pytest

from oracle import logmel as olm
from oracle.gen_golden import synth_cases


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filters_match_reference(golden_dir, n_mels):
    ref = np.load(f"{golden_dir}/mel_filters_{n_mels}.npy")
    got = olm.mel_filters(n_mels)
    assert got.dtype == np.float32 and got.shape == (n_mels, 201)
    assert float(np.abs(got - ref).max()) <= 1e-9


@pytest.mark.parametrize("n_mels", [80, 128])
def test_oracle_matches_reference_golden(golden_dir, n_mels):
    cases = synth_cases()
    cases["speech"] = [np.load(f"{golden_dir}/speech_pcm.npz")["pcm"]]
    for name, chunks in cases.items():
        g = np.load(f"{golden_dir}/logmel_{name}_{n_mels}.npz")
        assert list(g["n_samples"]) == [len(c) for c in chunks]
        got = olm.log_mel_chunks(chunks, n_mels)
        assert got.shape == (len(chunks), n_mels, 3000) and got.dtype == np.float32
        assert float(np.abs(got[..., g["frame_idx"]] - g["feats"]).max()) <= 1e-6, name
        full = olm.log_mel_full(chunks[-1], n_mels)
        assert full.shape[-1] == int(g["full_frames"][0])
        assert float(np.abs(full[..., g["full_idx"]] - g["full_last"]).max()) <= 1e-6, name


def test_c_restatement_matches_reference_golden(golden_dir):
    """oracle/logmel_ref.c (DFT by definition, double precision) vs the reference's float32 FFT:
    the difference is float32 round-off of the power spectrum only."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.join(os.path.dirname(__file__), "..", "oracle")
    subprocess.run(["make", "-C", here, "-s"], check=True)
    lib = C.CDLL(os.path.join(here, "_build", "liblogmel_ref.so"))
    lib.fw_oracle_logmel_full.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long]
    pcm = np.ascontiguousarray(np.load(f"{golden_dir}/speech_pcm.npz")["pcm"][:32000])
    for n_mels in (80, 128):
        nf = pcm.shape[0] // 160 + 1
        out = np.empty((n_mels, nf), np.float32)
        assert lib.fw_oracle_logmel_full(pcm.ctypes.data, pcm.shape[0], n_mels, out.ctypes.data, nf) == 0
        ref = olm.log_mel_full(pcm, n_mels)
        assert float(np.abs(out - ref).max()) < 1e-4
    empty = np.empty((80, 1), np.float32)
    assert lib.fw_oracle_logmel_full(None, 0, 80, empty.ctypes.data, 1) == 0
    assert abs(float(empty[0, 0]) + 1.5) < 1e-6


def test_edge_cases():
    # empty chunk: one frame of the floor value, then dropped -> all zeros after pad_or_trim
    out = olm.log_mel_chunks([np.zeros(0, np.float32)], 80)
    assert out.shape == (1, 80, 3000) and float(np.abs(out).max()) == 0.0
    full = olm.log_mel_full(np.zeros(0, np.float32), 80)
    assert full.shape == (80, 1) and abs(float(full[0, 0]) + 1.5) < 1e-6
    # > 30 s is trimmed to 3000 frames but normalised by the max over ALL frames
    rng = np.random.default_rng(0)
    x = (0.01 * rng.standard_normal(500000)).astype(np.float32)
    x[490000:] *= 100.0
    out = olm.log_mel_chunks([x], 80)
    alone = olm.log_mel_chunks([x[:480000]], 80)
    assert out.shape == (1, 80, 3000)
    assert float(np.abs(out - alone).max()) > 0.1   # the loud tail moved the clamp floor
