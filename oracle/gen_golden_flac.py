"""Golden fixture for the native FLAC decoder (row f-4), generated in the build container from the reference's own test
asset /root/reference/tests/data/jfk.flac (libFLAC-encoded: 44.1 kHz, stereo, 24 bit, 485 100 samples):

  1. the whole file is decoded by fw_flac_decode and must carry the encoder's MD5 signature (md5_status == 1): the decode is
     bit-exact with what libFLAC was given — that is what validates the decoder;
  2. the first 96 KiB of the file become tests/golden/flac_jfk_head.flac (the reference file itself is 1.1 MB and does not
     travel); the decoder returns the whole frames of a truncated stream, N samples;
  3. tests/golden/flac_jfk_head.json records N, the sha256 of the first N samples of the FULL (MD5-verified) decode, and the
     stream parameters.  tests/test_flac.py decodes the fixture anywhere (no /root/reference needed) and compares.

    python oracle/gen_golden_flac.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/tests/data/jfk.flac"
HEAD_BYTES = 96 * 1024


def decode(lib, data):
    from faster_whisper_amd import _lib
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rate, ch, bps, tot = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    _lib.check(lib.fw_flac_info(buf, len(data), C.byref(rate), C.byref(ch), C.byref(bps), C.byref(tot)))
    out = np.zeros((tot.value, ch.value), np.int32)
    n, md5 = C.c_int64(), C.c_int32()
    _lib.check(lib.fw_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), tot.value, C.byref(n), C.byref(md5)))
    return out[:n.value], md5.value, (rate.value, ch.value, bps.value, tot.value)


def main():
    from faster_whisper_amd import _lib
    lib = _lib.load()
    data = open(SRC, "rb").read()
    full, md5, params = decode(lib, data)
    assert md5 == 1 and full.shape[0] == params[3], (md5, full.shape, params)
    head = data[:HEAD_BYTES]
    part, md5h, _ = decode(lib, head)
    assert md5h == -1 and 0 < part.shape[0] < full.shape[0]
    n = part.shape[0]
    assert np.array_equal(part, full[:n])
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "flac_jfk_head.flac"), "wb") as f:
        f.write(head)
    meta = {"source": "faster-whisper tests/data/jfk.flac, first %d bytes" % HEAD_BYTES, "sample_rate": params[0],
            "channels": params[1], "bits_per_sample": params[2], "total_samples_in_streaminfo": params[3],
            "whole_frames_samples": n, "sha256_of_int32_le_samples": hashlib.sha256(full[:n].astype("<i4").tobytes()).hexdigest(),
            "full_file_md5_verified": True}
    with open(os.path.join(gold, "flac_jfk_head.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
