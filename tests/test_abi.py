"""The C-ABI library loads, exports every symbol include/*.h declare, and fails loudly
(no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = ""
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):      # fwamd.h (the boundary) + fwamd_test.h (hooks)
        if h.endswith(".h"):
            src += open(os.path.join(ROOT, "include", h)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fw_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from faster_whisper_amd import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding table covers the header
    assert sorted(_lib.SYMBOLS) == names
    assert lib.fw_abi_version() == 2


def test_struct_layouts_match_header():
    from faster_whisper_amd import _lib
    # fw_config: 20 scalar int32 + 1 + 8 + 1 + 128 int32
    assert C.sizeof(_lib.FwConfig) == 4 * (20 + 1 + 8 + 1 + 2 * _lib.FW_MAX_ALIGN_HEADS)
    assert C.sizeof(_lib.FwWeight) == 8 + 8 + 4 + 4 + 32
    assert _lib.FwGenOpts.suppress_tokens.offset % 8 == 0


def test_no_cpu_fallback(gpu_available):
    """without a HIP device model creation must raise; with one this test is vacuous"""
    from faster_whisper_amd import Whisper
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError) as e:
        Whisper("synthetic:micro", device="cuda")
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)
    with pytest.raises(ValueError):
        Whisper("synthetic:micro", device="cpu")


def test_config_presets():
    from faster_whisper_amd import get_config
    v3 = get_config("large-v3")
    assert (v3.n_mels, v3.d_model, v3.n_heads, v3.n_enc_layers, v3.n_dec_layers, v3.n_vocab) == (
        128, 1280, 20, 32, 32, 51866)
    assert v3.timestamp_begin == v3.no_timestamps + 1 == 50365
    en = get_config("tiny.en")
    # pinned by the reference's tests/test_tokenizer.py:110
    assert {en.sot, en.translate, en.transcribe, en.sot_lm, en.sot_prev, en.no_speech} == {
        50257, 50357, 50358, 50359, 50360, 50361}
    assert en.sot_sequence == [50257]
    d = get_config("distil-large-v3")
    assert d.n_dec_layers == 2 and d.n_enc_layers == 32
    c = v3.to_c()
    assert c.tok_eot == 50257 and c.n_langs == 100


def test_language_codes_cover_100():
    from faster_whisper_amd.backend import language_token_strings
    from faster_whisper_amd import get_config
    names = language_token_strings(get_config("large-v3"))
    assert len(names) == 100 and len(set(names)) == 100
    assert names[0] == "<|en|>" and names[-1] == "<|yue|>"
