"""Audio input: native RIFF/WAVE reader, polyphase resampler and the s16 round trip
of the reference's decode_audio (audio.py:19-76).  No GPU."""
import io
import struct
import wave

import numpy as np
import pytest

from faster_whisper_amd.audio import decode_audio, resample


def _wav16(path_or_buf, x, rate, channels=1):
    with wave.open(path_or_buf, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.asarray(x, dtype="<i2").tobytes())


def test_pcm16_mono_16k_is_exact(tmp_path):
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=16000, dtype=np.int16)
    p = str(tmp_path / "a.wav")
    _wav16(p, pcm, 16000)
    got = decode_audio(p)
    assert got.dtype == np.float32 and np.array_equal(got, pcm.astype(np.float32) / 32768.0)
    with open(p, "rb") as f:                       # file objects and bytes work too
        assert np.array_equal(decode_audio(f), got)
    assert np.array_equal(decode_audio(open(p, "rb").read()), got)


def test_stereo_downmix_and_split(tmp_path):
    left = (np.arange(800) % 100 * 100).astype(np.int16)
    right = (-(np.arange(800) % 50) * 200).astype(np.int16)
    p = str(tmp_path / "s.wav")
    _wav16(p, np.stack([left, right], 1).reshape(-1), 16000, channels=2)
    mono = decode_audio(p)
    want = np.rint((left.astype(np.float64) + right) / 2.0)     # half-sums: ties go to even like np.rint
    assert np.abs(mono * 32768.0 - want).max() <= 0.5 + 1e-6
    l, r = decode_audio(p, split_stereo=True)
    assert np.array_equal(l, left / np.float32(32768)) and np.array_equal(r, right / np.float32(32768))


def _wav_raw(fmt_code, bits, rate, channels, payload, extensible=False):
    block = channels * bits // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, 0, fmt_code,
                          b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    else:
        fmt = struct.pack("<HHIIHH", fmt_code, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + \
        b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_other_sample_formats():
    x = np.array([0.0, 0.5, -0.5, 0.25, -1.0, 0.999], dtype=np.float32)
    f32 = decode_audio(_wav_raw(3, 32, 16000, 1, x.astype("<f4").tobytes()))
    assert np.abs(f32 - x).max() <= 1.0 / 32768 + 1e-7 and f32[4] == -1.0       # s16 round trip clips / rounds
    f64 = decode_audio(_wav_raw(3, 64, 16000, 1, x.astype("<f8").tobytes(), extensible=True))
    assert np.array_equal(f64, f32)
    i24 = np.array([0, 1 << 22, -(1 << 22), 1 << 21, -(1 << 23), (1 << 23) - 256], dtype=np.int32)
    b24 = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in i24)
    assert np.abs(decode_audio(_wav_raw(1, 24, 16000, 1, b24)) - i24 / float(1 << 23)).max() <= 1.0 / 32768 + 1e-6
    u8 = np.array([128, 192, 64, 160, 0, 255], dtype=np.uint8)
    assert np.abs(decode_audio(_wav_raw(1, 8, 16000, 1, u8.tobytes())) - x).max() < 0.01
    with pytest.raises(ValueError):
        decode_audio(_wav_raw(2, 4, 16000, 1, b"\x00" * 8))                      # ADPCM: not read natively
    with pytest.raises(ValueError, match="FLAC"):
        decode_audio(b"fLaC" + b"\x00" * 64)                                     # FLAC is read natively: a corrupt one says so
    with pytest.raises(RuntimeError, match="PyAV"):
        decode_audio(b"OggS" + b"\x00" * 64)                                     # other containers need PyAV


@pytest.mark.parametrize("rate", [48000, 44100, 22050, 8000])
def test_resampling_preserves_in_band_tones(rate):
    t = np.arange(rate) / rate
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    y = resample(x, rate, 16000)
    assert len(y) == 16000
    to = np.arange(16000) / 16000
    ref = 0.5 * np.sin(2 * np.pi * 440 * to) + 0.2 * np.sin(2 * np.pi * 3000 * to)
    assert np.abs(y[300:-300] - ref[300:-300]).max() < 1e-4
    buf = io.BytesIO()
    _wav16(buf, np.rint(x * 32767).astype(np.int16), rate)
    dec = decode_audio(buf.getvalue())
    assert len(dec) == 16000 and np.abs(dec[300:-300] - ref[300:-300]).max() < 2e-4


def test_resampling_removes_out_of_band_energy():
    t = np.arange(48000) / 48000
    y = resample(np.sin(2 * np.pi * 10000 * t).astype(np.float32), 48000, 16000)      # above the new Nyquist
    assert np.abs(y[300:-300]).max() < 1e-3
    assert resample(np.zeros(0, np.float32), 48000, 16000).size == 0


@pytest.mark.parametrize("rate", [44100, 48000, 8000, 22050, 11025, 32000])
def test_polyphase_evaluation_equals_scipy_resample_poly(rate):
    """the resampler's polyphase indexing — phase alignment, output length ceil(n * up / down), zero-padded edges — against an
    independent implementation fed the SAME filter: scipy.signal.resample_poly(x, up, down, window=h / up) (scipy scales the
    taps it is given by `up`).  This pins the evaluation, not the filter design: libswresample's own filter
    (faster_whisper/audio.py:37-41 resamples through PyAV) stays unpinned without FFmpeg."""
    sig = pytest.importorskip("scipy.signal")
    from math import gcd
    rng = np.random.default_rng(rate)
    x = (rng.standard_normal(rate // 2 + 17) * 0.3).astype(np.float32)
    y = resample(x, rate, 16000)
    g = gcd(rate, 16000)
    up, down = 16000 // g, rate // g
    big, tpp, beta = max(up, down), 32, 9.0                       # resample()'s defaults
    half = tpp * big // 2
    t = np.arange(-half, half + 1, dtype=np.float64)
    cutoff = 0.5 / big
    h = 2 * cutoff * np.sinc(2 * cutoff * t) * np.kaiser(t.size, beta)
    h *= up / h.sum()
    z = sig.resample_poly(x.astype(np.float64), up, down, window=h / up)
    assert len(y) == len(z) == -(-len(x) * up // down)
    assert np.abs(y - z).max() < 1e-6
