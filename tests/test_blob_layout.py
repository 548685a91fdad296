"""Layout of the packed weight blob (host-side packer of the C-ABI library, no GPU): the MFMA-fragment-major
decoder linears, the LayerNorm folding and the header flag the engine dispatches on."""
import numpy as np

from faster_whisper_amd import _lib, get_config, synthetic_weights
from faster_whisper_amd.backend import pack_blob
from test_oracle_int8 import _parse_blob


def _frag(w):
    """reference permutation: W[n][k] -> [(n/16 * K/32 + k/32)][lane = 16*((k/8)%4) + n%16][k%8]"""
    n_rows, k_cols = w.shape
    ks = k_cols // 32
    out = np.empty(n_rows * k_cols, dtype=w.dtype)
    n, k = np.meshgrid(np.arange(n_rows), np.arange(k_cols), indexing="ij")
    off = (((n >> 4) * ks + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7)
    out[off.reshape(-1)] = w.reshape(-1)
    return out.reshape(n_rows, k_cols)


def _h(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16)


def test_fragment_major_decoder_linears(monkeypatch):
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=4)
    monkeypatch.setenv("FWAMD_DEC_GEMM", "frag")
    h, t = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    assert h.reserved & 1
    for name in ("dec.0.self.out.w", "dec.1.cross.out.w", "dec.0.ffn2.w"):
        assert np.array_equal(t[name], _frag(_h(w[name]))), name
    # LayerNorm-folded ones: (W * g) rounded to fp16, then permuted; s1 / cf stay plain
    g = _h(w["dec.1.ln3.g"]).astype(np.float32)
    folded = (_h(w["dec.1.ffn1.w"]).astype(np.float32) * g[None, :]).astype(np.float16)
    assert np.array_equal(t["dec.1.ffn1.wf"], _frag(folded))
    assert np.allclose(t["dec.1.ffn1.s1"], folded.astype(np.float64).sum(axis=1), rtol=0, atol=1e-3)
    # the "many rows" GEMM operands and the logits projection keep [N][K]
    assert np.array_equal(t["dec.0.cross.kv.w"], _h(w["dec.0.cross.kv.w"]))
    lg = (_h(w["dec.tok_emb"]).astype(np.float32) * _h(w["dec.ln.g"]).astype(np.float32)[None, :]).astype(np.float16)
    assert np.array_equal(t["dec.logits.wf"], lg)
    # the permutation is a bijection of the tile grid: every lane's 16 bytes are 8 consecutive k of one row
    x = np.arange(32 * 64, dtype=np.float32).reshape(32, 64)
    f = _frag(x).reshape(-1, 8)
    assert all((np.diff(r) == 1).all() and int(r[0]) % 8 == 0 for r in f)


def test_row_major_when_lds_form_is_selected(monkeypatch):
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=4)
    monkeypatch.setenv("FWAMD_DEC_GEMM", "lds")
    h, t = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    assert (h.reserved & 1) == 0
    assert np.array_equal(t["dec.0.self.out.w"], _h(w["dec.0.self.out.w"]))
    # int8 mode never uses the fragment-major form (its skinny GEMM stages through LDS)
    monkeypatch.setenv("FWAMD_DEC_GEMM", "frag")
    h8, _ = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_INT8_FLOAT16))
    assert (h8.reserved & 1) == 0


def test_int8_fragment_major_is_opt_in(monkeypatch):
    """FWAMD_DEC_GEMM_I8=frag (experiment): int8 decoder weights permuted for v_mfma_i32_16x16x64_i8 — element
    (n, k) at ((n/16 * K/64 + k/64) * 64 + 16*((k/16)%4) + n%16) * 16 + k%16 — flagged in header bit 1"""
    from oracle.whisper import OracleWhisper
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=6)
    monkeypatch.delenv("FWAMD_DEC_GEMM_I8", raising=False)
    h0, t0 = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_INT8_FLOAT16))
    assert (h0.reserved & 2) == 0
    monkeypatch.setenv("FWAMD_DEC_GEMM_I8", "frag")
    h1, t1 = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_INT8_FLOAT16))
    assert (h1.reserved & 2) == 2 and (h1.reserved & 1) == 0
    o = OracleWhisper(cfg, w, int8=True)
    for name in ("dec.0.self.qkv", "dec.1.ffn2", "dec.0.cross.out"):
        wq = o.q[name + ".w"][0].numpy().astype(np.int8)
        n_rows, k_cols = wq.shape
        n, k = np.meshgrid(np.arange(n_rows), np.arange(k_cols), indexing="ij")
        off = (((n >> 4) * (k_cols // 64) + (k >> 6)) * 64 + ((k >> 4) & 3) * 16 + (n & 15)) * 16 + (k & 15)
        want = np.empty(n_rows * k_cols, np.int8)
        want[off.reshape(-1)] = wq.reshape(-1)
        assert np.array_equal(t1[name + ".wq"].reshape(-1), want), name
        assert np.array_equal(t0[name + ".wq"], wq)                       # default: row-major
        assert np.array_equal(t1[name + ".ws"], t0[name + ".ws"])
    # encoder, cross-K/V and logits weights are not touched
    for name in ("enc.0.ffn1.wq", "dec.0.cross.kv.wq", "dec.logits.wq"):
        assert np.array_equal(t1[name], t0[name])
    # fp16 packing ignores the int8 knob
    h16, _ = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    assert (h16.reserved & 2) == 0
