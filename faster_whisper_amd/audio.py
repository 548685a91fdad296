"""Audio input of the pipelines: file -> 16 kHz mono float32.

Mirrors `faster_whisper/audio.py:19-76` (`decode_audio(input_file, sampling_rate=16000, split_stereo=False)`):
the reference decodes with PyAV (bundled FFmpeg), resamples to signed 16-bit at the target rate and returns
`int16 / 32768` as float32 (mono, or a (left, right) pair with `split_stereo`).  Here:
  * RIFF/WAVE files (PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE) are read natively;
  * FLAC streams are decoded natively by libfwamd (`fw_flac_decode`, csrc/flac_host.cpp: every frame CRC-checked, the
    decoded PCM checked against the MD5 signature the encoder stored — the reference's own tests/data/jfk.flac decodes
    bit-exactly);
  * other containers (MP3, AAC, Ogg ...) are delegated to PyAV when it is importable, and fail loudly otherwise;
  * rate conversion is a Kaiser-windowed sinc polyphase filter in numpy (libswresample is not available, so the
    resampled waveform is not bit-identical to the reference's — the s16 quantisation step and the interface are).
`pad_or_trim` (audio.py:111-123) lives in transcribe.py.
"""
import io
import os
import struct
from math import gcd
from typing import BinaryIO, Tuple, Union

import numpy as np


def _read_wav(data: bytes) -> Tuple[np.ndarray, int]:
    """-> (float32 [frames, channels] in [-1, 1), sample rate)"""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if code == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first word
                code = struct.unpack("<H", body[24:26])[0]
            fmt = (code, channels, rate, bits)
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("WAVE file without 'fmt ' or 'data' chunk")
    code, channels, rate, bits = fmt
    if channels < 1:
        raise ValueError("WAVE file with no channels")
    if code == 1:          # integer PCM
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (np.frombuffer(pcm[:len(pcm) // 4 * 4], dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"unsupported PCM bit depth {bits}")
    elif code == 3:        # IEEE float
        if bits not in (32, 64):
            raise ValueError(f"unsupported float bit depth {bits}")
        x = np.frombuffer(pcm[:len(pcm) // (bits // 8) * (bits // 8)], dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE format code {code} (only PCM and IEEE float are read natively)")
    n = len(x) // channels
    return x[:n * channels].reshape(n, channels), rate


def max_decoded_seconds() -> float:
    """ceiling on the audio one file may decode to (FWAMD_MAX_AUDIO_SECONDS, default 48 h): a FLAC stream of CONSTANT
    subframes codes 65 535 samples per channel in ~11 bytes, so a small crafted file could otherwise ask for tens of GB"""
    return float(os.environ.get("FWAMD_MAX_AUDIO_SECONDS", 48 * 3600))


def _read_flac(data: bytes) -> Tuple[np.ndarray, int]:
    """-> (float32 [frames, channels] in [-1, 1), sample rate); raises ValueError on a corrupt stream (CRC / MD5) and on a
    stream that decodes to more than max_decoded_seconds() of audio"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rate, ch, bps, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    _lib.check(lib.fw_flac_info(buf, len(data), C.byref(rate), C.byref(ch), C.byref(bps), C.byref(total)))
    # Output sizing.  STREAMINFO's total is untrusted input (36 bits) and 0 for a streamed encode, and there is no useful
    # lower bound on bytes per sample (a CONSTANT subframe codes a whole block of up to 65 535 samples in a few bytes):
    # start from what the data plausibly holds and grow on "too small" up to the format's own ceiling for this many
    # bytes (a frame is at least 11 bytes: header, one subframe byte per channel, CRC-16).
    hard_cap = (len(data) // 11 + 1) * 65535
    limit = int(max_decoded_seconds() * max(1, rate.value))
    if 0 < total.value <= hard_cap and total.value > limit:
        raise ValueError(f"FLAC: the stream announces {total.value / max(1, rate.value):.0f} s of audio, more than the "
                         f"{max_decoded_seconds():.0f} s this front end decodes (FWAMD_MAX_AUDIO_SECONDS)")
    want = min(total.value, hard_cap) if total.value > 0 else hard_cap
    bounded = want > limit           # (only an unannounced / implausible length gets here with want > limit)
    want = min(want, limit)
    cap = max(1, min(want, 16 * len(data)))
    n, md5 = C.c_int64(), C.c_int32()
    while True:
        out = np.zeros((cap, ch.value), dtype=np.int32)
        rc = lib.fw_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(md5))
        if rc == _lib.FW_ENOSPC and cap < want:
            cap = min(want, cap * 4)
            continue
        if rc == _lib.FW_ENOSPC and bounded:
            raise ValueError(f"FLAC: the stream decodes to more than {max_decoded_seconds():.0f} s of audio "
                             "(FWAMD_MAX_AUDIO_SECONDS): refusing to allocate for it")
        _lib.check(rc)
        break
    if 0 < total.value != n.value:
        import warnings
        warnings.warn(f"FLAC: {n.value} of the {total.value} samples STREAMINFO announces were decoded "
                      "(truncated stream or lost frame sync)", RuntimeWarning)
    if md5.value == 0:
        raise ValueError("FLAC: the decoded audio does not carry the MD5 signature stored in the stream (corrupt file)")
    x = (out[:n.value].astype(np.float64) / float(1 << (bps.value - 1))).astype(np.float32)
    return x, rate.value


def resample(x: np.ndarray, rate_in: int, rate_out: int, taps_per_phase: int = 32, beta: float = 9.0) -> np.ndarray:
    """Polyphase Kaiser-windowed-sinc rate conversion of a 1-D float signal (zero phase, unit pass-band gain):
    conceptually zero-stuff by `up`, low-pass at min(rate_in, rate_out) / 2, keep every `down`-th sample —
    evaluated per output phase so that only the taps that meet a real input sample are multiplied."""
    x = np.asarray(x, dtype=np.float32)
    if rate_in == rate_out or x.size == 0:
        return x
    g = gcd(rate_in, rate_out)
    up, down = rate_out // g, rate_in // g
    big = max(up, down)
    half = taps_per_phase * big // 2
    t = np.arange(-half, half + 1, dtype=np.float64)
    cutoff = 0.5 / big                                # cycles per sample of the zero-stuffed stream
    h = 2 * cutoff * np.sinc(2 * cutoff * t) * np.kaiser(t.size, beta)
    h *= up / h.sum()
    n_out = -(-x.size * up // down)
    pad = half // up + 2
    xp = np.concatenate([np.zeros(pad), x.astype(np.float64), np.zeros(pad + down)])
    out = np.empty(n_out, dtype=np.float64)
    # output m sits at c = m*down on the zero-stuffed grid and sees inputs k with |k*up - c| <= half:
    # y[m] = sum_k h[half + k*up - c] x[k].  Outputs m0, m0 + up, m0 + 2 up, ... share their tap subset.
    for m0 in range(min(up, n_out)):
        c0 = m0 * down
        j0 = (half - c0) % up                         # first tap that lands on an input sample
        k0 = (c0 - half + j0) // up                   # that input's index (may be negative: zero padding)
        coef = h[j0::up]
        rows = np.arange(m0, n_out, up)
        for lo in range(0, rows.size, 1 << 15):
            r = rows[lo:lo + (1 << 15)]
            starts = pad + k0 + (r - m0) // up * down
            win = np.lib.stride_tricks.sliding_window_view(xp, coef.size)
            valid = np.minimum(starts, win.shape[0] - 1)
            out[r] = win[valid] @ coef
    return out.astype(np.float32)


def _to_s16_float(x: np.ndarray) -> np.ndarray:
    """the reference hands back int16 / 32768 (audio.py:62-66): same quantisation"""
    q = np.clip(np.rint(np.asarray(x, dtype=np.float64) * 32768.0), -32768, 32767)
    return (q / 32768.0).astype(np.float32)


def _decode_with_pyav(input_file, sampling_rate: int, split_stereo: bool):
    import av
    resampler = av.audio.resampler.AudioResampler(format="s16", layout="stereo" if split_stereo else "mono",
                                                  rate=sampling_rate)
    chunks = []
    with av.open(input_file, mode="r", metadata_errors="ignore") as container:
        for frame in container.decode(audio=0):
            frame.pts = None
            for out in resampler.resample(frame):
                chunks.append(out.to_ndarray().reshape(-1))
        for out in resampler.resample(None):
            chunks.append(out.to_ndarray().reshape(-1))
    audio = (np.concatenate(chunks) if chunks else np.zeros(0, np.int16)).astype(np.float32) / 32768.0
    return (audio[0::2], audio[1::2]) if split_stereo else audio


def decode_audio(input_file: Union[str, BinaryIO], sampling_rate: int = 16000, split_stereo: bool = False):
    """-> float32 waveform at `sampling_rate` (mono), or (left, right) when split_stereo"""
    if isinstance(input_file, (str, bytes)) and not isinstance(input_file, bytes):
        with open(input_file, "rb") as f:
            data = f.read()
    elif isinstance(input_file, bytes):
        data = input_file
    else:
        data = input_file.read()
    native = None
    if data[:4] == b"RIFF" and data[8:12] == b"WAVE":
        native = _read_wav(data)
    elif data[:4] == b"fLaC" or (data[:3] == b"ID3" and b"fLaC" in data[:1 << 20]):
        native = _read_flac(data)
    if native is not None:
        x, rate = native
        if split_stereo:
            if x.shape[1] < 2:
                x = np.repeat(x[:, :1], 2, axis=1)
            left, right = (resample(x[:, c], rate, sampling_rate) for c in (0, 1))
            return _to_s16_float(left), _to_s16_float(right)
        mono = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
        return _to_s16_float(resample(mono, rate, sampling_rate))
    try:
        import av  # noqa: F401
    except ImportError as e:
        raise RuntimeError("only RIFF/WAVE and FLAC are decoded natively; other containers need the PyAV package "
                           "(the reference's decoder), which is not installed") from e
    return _decode_with_pyav(io.BytesIO(data), sampling_rate, split_stereo)
