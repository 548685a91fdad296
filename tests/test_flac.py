"""Native FLAC decoding (reference: faster_whisper/audio.py:19-76 decodes through PyAV).

Two anchors, no GPU:
  * tests/golden/flac_jfk_head.flac — the first 96 KiB of the reference's own test asset (libFLAC-encoded, 44.1 kHz stereo
    24 bit, LPC subframes, Rice partitions, decorrelated stereo).  oracle/gen_golden_flac.py decoded the WHOLE file in the
    build container and required the encoder's MD5 signature to match (bit-exact decode), then recorded the hash of the
    samples of the fixture's whole frames.  Here the fixture must decode to exactly those samples.
  * streams written by the small encoder below (restated from the published format, independent of the decoder's code
    path by construction: it WRITES bits): mono / stereo / 3 channels, 8 / 16 / 24 bit, CONSTANT, VERBATIM and FIXED
    subframes of every order, Rice and Rice2 parameters, escaped partitions, several partitions, wasted bits,
    left-side / right-side / mid-side, odd block sizes, an unknown stream length, an ID3v2 tag in front.  Every stream carries
    the MD5 of its PCM: the decoder must return the PCM and report the signature as matching; flipped payload bytes must
    be caught by the frame CRC."""
import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from faster_whisper_amd import _lib


def _decode(data: bytes, cap=None):
    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rate, ch, bps, tot = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    _lib.check(lib.fw_flac_info(buf, len(data), C.byref(rate), C.byref(ch), C.byref(bps), C.byref(tot)))
    cap = cap if cap is not None else (tot.value if tot.value > 0 else 8 * len(data))
    out = np.zeros((cap, ch.value), np.int32)
    n, md5 = C.c_int64(), C.c_int32()
    _lib.check(lib.fw_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(md5)))
    return out[:n.value], md5.value, (rate.value, ch.value, bps.value, tot.value)


def test_reference_asset_head_decodes_to_the_md5_verified_samples():
    with open(os.path.join(GOLDEN, "flac_jfk_head.json")) as f:
        meta = json.load(f)
    data = open(os.path.join(GOLDEN, "flac_jfk_head.flac"), "rb").read()
    pcm, md5, (rate, ch, bps, tot) = _decode(data)
    assert (rate, ch, bps, tot) == (meta["sample_rate"], meta["channels"], meta["bits_per_sample"],
                                    meta["total_samples_in_streaminfo"])
    assert md5 == -1                                   # a truncated stream: nothing to compare the signature with
    assert pcm.shape == (meta["whole_frames_samples"], ch)
    assert hashlib.sha256(pcm.astype("<i4").tobytes()).hexdigest() == meta["sha256_of_int32_le_samples"]


# ---------------------------------------------------------------------------------------------------------------------
# a minimal FLAC writer
# ---------------------------------------------------------------------------------------------------------------------
class _W:
    def __init__(self):
        self.bits = []

    def u(self, v, k):
        assert 0 <= v < (1 << k) or k == 0, (v, k)
        self.bits.extend((v >> (k - 1 - i)) & 1 for i in range(k))

    def s(self, v, k):
        self.u(v & ((1 << k) - 1), k)

    def unary(self, q):
        self.bits.extend([0] * q + [1])

    def pad(self):
        self.bits.extend([0] * (-len(self.bits) % 8))

    def bytes(self):
        assert len(self.bits) % 8 == 0
        b = np.packbits(np.array(self.bits, dtype=np.uint8))
        return b.tobytes()


def _crc(data, poly, width):
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _subframe(w, x, bps, kind, wasted=0, rice=(0, 3, 0), escape=False):
    """x: python ints of one channel.  kind: 'const' | 'verbatim' | ('fixed', order).  rice = (method, k, partition order)"""
    w.u(0, 1)
    code = 0 if kind == "const" else 1 if kind == "verbatim" else 8 + kind[1]
    w.u(code, 6)
    if wasted:
        w.u(1, 1)
        w.unary(wasted - 1)
        x = [v >> wasted for v in x]
        bps -= wasted
    else:
        w.u(0, 1)
    if kind == "const":
        w.s(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            w.s(v, bps)
    else:
        order = kind[1]
        for v in x[:order]:
            w.s(v, bps)
        res = [x[i] - sum(c * x[i - 1 - j] for j, c in enumerate(_FIXED[order])) for i in range(order, len(x))]
        method, k, porder = rice
        w.u(method, 2)
        w.u(porder, 4)
        n_part = 1 << porder
        per = len(x) >> porder
        pos = 0
        for p in range(n_part):
            cnt = per - (order if p == 0 else 0)
            part = res[pos:pos + cnt]
            pos += cnt
            if escape and p % 2 == 0:
                w.u(15 if method == 0 else 31, 4 if method == 0 else 5)
                nb = max(1, max((abs(v).bit_length() + 1 for v in part), default=1))
                w.u(nb, 5)
                for v in part:
                    w.s(v, nb)
            else:
                w.u(k, 4 if method == 0 else 5)
                for v in part:
                    u = (v << 1) if v >= 0 else ((-v) << 1) - 1
                    w.unary(u >> k)
                    w.u(u & ((1 << k) - 1), k)
        assert pos == len(res)


def _frame(index, chans, bps, sub, stereo=None, block_code=None):
    """chans: list of per-channel int lists of one block.  sub: per-channel _subframe keyword dicts.
    stereo: None | 'ls' | 'rs' | 'ms'"""
    n = len(chans[0])
    w = _W()
    w.u(0b11111111111110, 14)
    w.u(0, 1)
    w.u(0, 1)                                          # fixed block size stream: frames are numbered
    if block_code is None:
        block_code = 6 if n <= 256 else 7
    w.u(block_code, 4)
    w.u(0, 4)                                          # sample rate: from STREAMINFO
    w.u({None: len(chans) - 1, "ls": 8, "rs": 9, "ms": 10}[stereo], 4)
    w.u({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3)
    w.u(0, 1)
    assert index < 0x800
    if index < 0x80:
        w.u(index, 8)
    else:
        w.u(0xC0 | (index >> 6), 8)
        w.u(0x80 | (index & 0x3F), 8)
    if block_code == 6:
        w.u(n - 1, 8)
    elif block_code == 7:
        w.u(n - 1, 16)
    head = w.bytes()
    w.u(_crc(head, 0x07, 8), 8)
    coded = [list(c) for c in chans]
    extra = [0] * len(chans)
    if stereo == "ls":
        coded[1] = [a - b for a, b in zip(chans[0], chans[1])]
        extra[1] = 1
    elif stereo == "rs":
        coded[0] = [a - b for a, b in zip(chans[0], chans[1])]
        coded[1] = list(chans[1])
        extra[0] = 1
    elif stereo == "ms":
        coded[0] = [(a + b) >> 1 for a, b in zip(chans[0], chans[1])]
        coded[1] = [a - b for a, b in zip(chans[0], chans[1])]
        extra[1] = 1
    for c, x in enumerate(coded):
        _subframe(w, x, bps + extra[c], **sub[c])
    w.pad()
    body = w.bytes()
    return body + struct.pack(">H", _crc(body, 0x8005, 16))


def _stream(pcm, bps, rate, blocks, known_length=True, id3=False):
    """pcm: int array [n][channels]; blocks: list of (block size, per-channel subframe dicts, stereo mode, block code)"""
    n, ch = pcm.shape
    nbytes = (bps + 7) // 8
    md5 = hashlib.md5(b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in pcm.reshape(-1))).digest()
    frames, pos = [], 0
    for i, (bs, sub, stereo, code) in enumerate(blocks):
        frames.append(_frame(i, [[int(v) for v in pcm[pos:pos + bs, c]] for c in range(ch)], bps, sub, stereo, code))
        pos += bs
    assert pos == n
    w = _W()
    w.u(max(b[0] for b in blocks), 16)
    w.u(max(b[0] for b in blocks), 16)
    w.u(0, 24)
    w.u(0, 24)
    w.u(rate, 20)
    w.u(ch - 1, 3)
    w.u(bps - 1, 5)
    w.u(n if known_length else 0, 36)
    info = w.bytes() + md5
    out = b"fLaC" + bytes([0x00]) + len(info).to_bytes(3, "big") + info                      # STREAMINFO
    out += bytes([0x80 | 1]) + (7).to_bytes(3, "big") + b"\0" * 7                            # a PADDING block, last
    out += b"".join(frames)
    if id3:
        out = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 12]) + b"\0" * 12 + out
    return out


def _signal(n, ch, bps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) * 0.4
    x = np.stack([amp * np.sin(2 * np.pi * (0.003 + 0.002 * c) * t + c) + amp * 0.02 * rng.standard_normal(n)
                  for c in range(ch)], axis=1)
    return np.clip(np.rint(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


CASES = {
    "mono 16 bit, fixed orders 0-4, rice": dict(
        ch=1, bps=16, blocks=[(192, [dict(kind=("fixed", o), rice=(0, 6, 0))], None, 1) for o in range(5)]),
    "stereo 16 bit, independent / left-side / right-side / mid-side, partitions, rice2": dict(
        ch=2, bps=16, blocks=[(256, [dict(kind=("fixed", 2), rice=(1, 5, 3)), dict(kind=("fixed", 1), rice=(0, 7, 2))], m, 6)
                              for m in (None, "ls", "rs", "ms")]),
    "stereo 24 bit, escaped partitions, verbatim, 16-bit block size field": dict(
        ch=2, bps=24, blocks=[(1000, [dict(kind=("fixed", 3), rice=(0, 9, 1), escape=True), dict(kind="verbatim")], "ms", 7),
                              (333, [dict(kind=("fixed", 4), rice=(1, 10, 0)), dict(kind=("fixed", 0), rice=(1, 17, 0))], None, 7)]),
    "3 channels 8 bit, verbatim + fixed": dict(
        ch=3, bps=8, blocks=[(100, [dict(kind="verbatim"), dict(kind=("fixed", 1), rice=(0, 2, 0)),
                                    dict(kind=("fixed", 2), rice=(0, 3, 0))], None, 6)] * 3),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("known_length,id3", [(True, False), (False, True)])
def test_written_streams_decode_bit_exactly(name, known_length, id3):
    c = CASES[name]
    n = sum(b[0] for b in c["blocks"])
    pcm = _signal(n, c["ch"], c["bps"], seed=len(name))
    data = _stream(pcm, c["bps"], 22050, c["blocks"], known_length=known_length, id3=id3)
    got, md5, (rate, ch, bps, tot) = _decode(data)
    assert (rate, ch, bps) == (22050, c["ch"], c["bps"]) and tot == (n if known_length else 0)
    assert got.shape == pcm.shape and np.array_equal(got, pcm)
    assert md5 == 1                                    # the signature in STREAMINFO is the PCM's


def test_constant_subframes_and_wasted_bits():
    n = 192
    pcm = np.zeros((2 * n, 2), np.int64)
    pcm[:n, 0] = 1234
    pcm[:n, 1] = (_signal(n, 1, 12, 3)[:, 0]) << 4     # 16-bit samples whose low 4 bits are zero
    pcm[n:, 0] = -77
    pcm[n:, 1] = (_signal(n, 1, 10, 4)[:, 0]) << 6
    blocks = [(n, [dict(kind="const"), dict(kind=("fixed", 2), rice=(0, 5, 0), wasted=4)], None, 1),
              (n, [dict(kind="const"), dict(kind="verbatim", wasted=6)], None, 1)]
    got, md5, _ = _decode(_stream(pcm, 16, 16000, blocks))
    assert np.array_equal(got, pcm) and md5 == 1


def test_corruption_is_caught():
    c = CASES["mono 16 bit, fixed orders 0-4, rice"]
    pcm = _signal(sum(b[0] for b in c["blocks"]), 1, 16, seed=9)
    data = bytearray(_stream(pcm, 16, 16000, c["blocks"]))
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x10                         # a payload bit: the frame's CRC-16 (or its parse) must object
    with pytest.raises(ValueError, match="FLAC"):
        _decode(bytes(bad))
    sig = bytearray(data)
    sig[4 + 4 + 18] ^= 0xFF                            # the stored MD5 itself: decode succeeds, signature does not match
    got, md5, _ = _decode(bytes(sig))
    assert np.array_equal(got, pcm) and md5 == 0
    with pytest.raises(ValueError, match="fLaC"):
        _decode(b"RIFFxxxxWAVE" + bytes(64))
    with pytest.raises(ValueError, match="too small"):
        _decode(bytes(data), cap=10)


def test_decode_audio_reads_flac_natively():
    """decode_audio (audio.py:19-76 in the reference) on FLAC input: float32 at 16 kHz, mono mix or split stereo, s16 grid"""
    import io
    from faster_whisper_amd.audio import decode_audio
    n = 4096
    pcm = _signal(n, 2, 16, seed=1)
    blocks = [(1024, [dict(kind=("fixed", 2), rice=(0, 8, 2)), dict(kind=("fixed", 2), rice=(0, 8, 2))], "ms", 7)] * 4
    data = _stream(pcm, 16, 16000, blocks)
    mono = decode_audio(io.BytesIO(data))
    assert mono.dtype == np.float32 and mono.shape == (n,)
    want = np.clip(np.rint(pcm.mean(axis=1)), -32768, 32767) / 32768.0
    assert np.abs(mono - want).max() <= 1.0 / 32768.0 + 1e-7
    left, right = decode_audio(data, split_stereo=True)
    assert np.array_equal(left, (pcm[:, 0] / 32768.0).astype(np.float32))
    assert np.array_equal(right, (pcm[:, 1] / 32768.0).astype(np.float32))
    # another rate: resampled to 16 kHz (length only; the resampler has its own tests in test_audio.py)
    data8 = _stream(pcm, 16, 8000, blocks)
    assert decode_audio(data8).shape == (2 * n,)
    # the reference's asset, fixture form: 36 864 samples at 44.1 kHz -> 13 375 at 16 kHz
    head = open(os.path.join(GOLDEN, "flac_jfk_head.flac"), "rb").read()
    a = decode_audio(head)
    assert a.shape == (-(-36864 * 160 // 441),) and np.isfinite(a).all() and np.abs(a).max() <= 1.0


def test_streamed_digital_silence_and_truncation_warning():
    """A streamed encode (STREAMINFO total = 0) of digital silence: CONSTANT subframes code 4 608 samples per ~14-byte frame,
    far more than any bytes-per-sample bound allows for — the reader grows its output instead of failing with "too small".
    A stream cut short of its announced length is returned with a warning (FFmpeg, the reference's decoder, also keeps going)."""
    import warnings
    from faster_whisper_amd.audio import _read_flac
    bs, nb = 4608, 40
    pcm = np.zeros((bs * nb, 1), np.int64)
    blocks = [(bs, [dict(kind="const")], None, 5)] * nb            # block code 5 = 4 608 samples, 8-bit-free header
    data = _stream(pcm, 16, 16000, blocks, known_length=False)
    assert 16 * len(data) < bs * nb                                # the first guess IS too small
    x, rate = _read_flac(data)
    assert rate == 16000 and x.shape == (bs * nb, 1) and not x.any()
    known = _stream(pcm, 16, 16000, blocks, known_length=True)
    cut = known[:len(known) - 3 * (len(known) - len(known[:known.index(b"\xff\xf8")])) // nb]   # drop the last frames
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        y, _ = _read_flac(cut)
    assert 0 < y.shape[0] < bs * nb and y.shape[0] % bs == 0
    assert any("samples STREAMINFO announces" in str(w.message) for w in rec)


def test_decompression_bomb_is_refused(monkeypatch):
    """ADVICE (round 5): a small stream of CONSTANT frames may decode to hours of audio.  Above FWAMD_MAX_AUDIO_SECONDS the
    reader raises a clear ValueError — for an announced length before any allocation, for a streamed encode (total = 0) once
    the capped buffer has proved too small — instead of growing towards the format's ceiling; the C side reports the small
    buffer with its own status (FW_ENOSPC), not by message text."""
    from faster_whisper_amd import _lib
    from faster_whisper_amd.audio import _read_flac
    bs, nb = 4608, 40
    pcm = np.zeros((bs * nb, 1), np.int64)
    blocks = [(bs, [dict(kind="const")], None, 5)] * nb
    lib = _lib.load()
    for known in (True, False):
        data = _stream(pcm, 16, 16000, blocks, known_length=known)
        monkeypatch.setenv("FWAMD_MAX_AUDIO_SECONDS", "5")          # the stream holds 11.5 s
        with pytest.raises(ValueError, match="FWAMD_MAX_AUDIO_SECONDS"):
            _read_flac(data)
        monkeypatch.setenv("FWAMD_MAX_AUDIO_SECONDS", "12")
        x, _ = _read_flac(data)
        assert x.shape == (bs * nb, 1)
    import ctypes as C
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    out = np.zeros((10, 1), np.int32)
    n, md5 = C.c_int64(), C.c_int32()
    assert lib.fw_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), 10, C.byref(n), C.byref(md5)) == _lib.FW_ENOSPC
