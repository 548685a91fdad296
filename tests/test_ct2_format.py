"""CTranslate2 `model.bin` reader: binary round trip, name mapping, int8
de-quantisation, loud failure on malformed files.  [CT2-ext] the layout itself is restated from
the published converter and cannot be checked against a real checkpoint offline."""
import os
import struct

import numpy as np
import pytest

from faster_whisper_amd import get_config, synthetic_weights
from faster_whisper_amd.backend import load_model_dir
from faster_whisper_amd.ct2_format import (load_ct2_model_dir, name_map, read_model_bin, save_ct2_model_dir,
                                           write_model_bin)
from faster_whisper_amd.weights import weight_shapes


def test_name_map_is_a_bijection_onto_the_engine_names():
    cfg = get_config("tiny.en")
    m = name_map(cfg.n_enc_layers, cfg.n_dec_layers)
    assert sorted(m.values()) == sorted(weight_shapes(cfg).keys())
    assert m["decoder/layer_3/attention/linear_1/weight"] == "dec.3.cross.kv.w"
    assert m["encoder/layer_0/self_attention/linear_0/weight"] == "enc.0.attn.qkv.w"


def test_roundtrip_through_a_ct2_directory(tmp_path):
    cfg = get_config("micro")
    cfg.alignment_heads = [(1, 0), (1, 1)]
    w = synthetic_weights(cfg, seed=5)
    d = str(tmp_path / "ct2model")
    save_ct2_model_dir(d, cfg, w)
    spec, rev, variables, aliases = read_model_bin(os.path.join(d, "model.bin"))
    assert spec == "WhisperSpec" and rev == 3 and aliases == {"decoder/projection/weight": "decoder/embeddings/weight"}
    assert variables["encoder/conv1/weight"].dtype == np.float16
    # micro's vocabulary is not a Whisper vocabulary: the loader must refuse it ...
    with pytest.raises(ValueError):
        load_ct2_model_dir(d)
    # ... but a Whisper-sized vocabulary loads and reproduces geometry + every tensor
    cfg2 = get_config("tiny.en")
    cfg2.n_enc_layers = cfg2.n_dec_layers = 1
    w2 = synthetic_weights(cfg2, seed=6)
    d2 = str(tmp_path / "tiny1")
    save_ct2_model_dir(d2, cfg2, w2)
    got_cfg, got_w = load_model_dir(d2)     # the backend's loader dispatches on model.bin
    assert (got_cfg.d_model, got_cfg.n_heads, got_cfg.n_mels, got_cfg.n_enc_layers, got_cfg.n_dec_layers,
            got_cfg.n_vocab, got_cfg.is_multilingual) == (384, 6, 80, 1, 1, 51864, False)
    assert got_cfg.eot == 50256 and got_cfg.timestamp_begin == 50363 and got_cfg.suppress_begin == (220, 50256)
    assert set(got_w) == set(w2)
    for k in w2:
        assert np.array_equal(got_w[k], w2[k]), k


def test_int8_variables_are_dequantised(tmp_path):
    rng = np.random.default_rng(0)
    wq = rng.integers(-127, 128, size=(4, 8), dtype=np.int8)
    scale = np.array([127.0, 63.5, 12.7, 1.0], dtype=np.float32)
    from faster_whisper_amd.ct2_format import _dequant
    out = _dequant({"x/weight": wq, "x/weight_scale": scale}, "x/weight")
    assert out.dtype == np.float32 and np.allclose(out, wq.astype(np.float32) / scale[:, None])
    with pytest.raises(ValueError):
        _dequant({"x/weight": wq}, "x/weight")


def test_malformed_files_fail_loudly(tmp_path):
    p = str(tmp_path / "model.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<I", 99999))
    with pytest.raises(ValueError):
        read_model_bin(p)
    write_model_bin(p, {"a": np.zeros((2, 3), np.float32)})
    raw = bytearray(open(p, "rb").read())
    raw[-30] ^= 0xFF                      # corrupt the payload region / counts
    open(p, "wb").write(bytes(raw[:-8]))  # and truncate
    with pytest.raises((ValueError, struct.error)):
        read_model_bin(p)
