"""Layout of the packed weight blob (host-side packer of the C-ABI library, no GPU): the MFMA-fragment-major
decoder linears and vocabulary projection (fp16 and int8), the LayerNorm folding and the header's layout
generation."""
import numpy as np

from conftest import frag_perm, frag_unperm
from faster_whisper_amd import _lib, get_config, synthetic_weights
from faster_whisper_amd.backend import pack_blob
from test_oracle_int8 import _parse_blob


def _h(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16)


def test_fragment_major_decoder_linears(monkeypatch):
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=4)
    h0, t0 = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    # the default blob carries the folded forms only (layout generation 6): no plain qkv / cross.q / ffn1 weights, no
    # second copy of the tied embedding (+1 GB at large-v3 for a default-off diagnostic otherwise)
    assert h0.reserved == 6
    assert "dec.0.self.qkv.wf" in t0 and "dec.0.self.qkv.w" not in t0 and "dec.logits.wp" not in t0 and "dec.1.ln3.g" not in t0
    monkeypatch.setenv("FWAMD_PACK_PLAIN", "1")
    h, t = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    assert h.reserved == 6 and all(np.array_equal(t[k], t0[k]) for k in t0)
    for name in ("dec.0.self.out.w", "dec.1.cross.out.w", "dec.0.ffn2.w"):
        assert np.array_equal(t[name], frag_perm(_h(w[name]))), name
    # LayerNorm-folded ones: (W * g) rounded to fp16, then permuted; s1 / cf stay plain
    g = _h(w["dec.1.ln3.g"]).astype(np.float32)
    folded = (_h(w["dec.1.ffn1.w"]).astype(np.float32) * g[None, :]).astype(np.float16)
    assert np.array_equal(t["dec.1.ffn1.wf"], frag_perm(folded))
    assert np.allclose(t["dec.1.ffn1.s1"], folded.astype(np.float64).sum(axis=1), rtol=0, atol=1e-3)
    # the "many rows" GEMM operands keep [N][K]
    assert np.array_equal(t["dec.0.cross.kv.w"], _h(w["dec.0.cross.kv.w"]))
    # the vocabulary projection: folded final LayerNorm, N padded to whole 16-column tiles with zero rows
    lg = (_h(w["dec.tok_emb"]).astype(np.float32) * _h(w["dec.ln.g"]).astype(np.float32)[None, :]).astype(np.float16)
    V = cfg.n_vocab
    VP = (V + 15) // 16 * 16
    assert t["dec.logits.wf"].shape == (VP, cfg.d_model)
    un = frag_unperm(t["dec.logits.wf"])
    assert np.array_equal(un[:V], lg) and not un[V:].any()
    assert t["dec.logits.s1"].shape == (V,)
    # packed with FWAMD_PACK_PLAIN / FWAMD_LN_UNFOLD: the explicit-LayerNorm forms travel too — plain weights fragment-major, LayerNorm gain / bias,
    # the tied embedding as a fragment-major projection (zero-padded like the folded one)
    for name in ("dec.0.self.qkv.w", "dec.1.cross.q.w", "dec.1.ffn1.w"):
        assert np.array_equal(t[name], frag_perm(_h(w[name]))), name
    assert np.array_equal(t["dec.1.ln3.g"], _h(w["dec.1.ln3.g"])) and np.array_equal(t["dec.ln.b"], _h(w["dec.ln.b"]))
    up = frag_unperm(t["dec.logits.wp"])
    assert np.array_equal(up[:V], _h(w["dec.tok_emb"])) and not up[V:].any()
    assert np.array_equal(t["dec.tok_emb"], _h(w["dec.tok_emb"]))      # the embedding lookup keeps its row-major copy
    # the permutation is a bijection of the tile grid: every lane's 16 bytes are 8 consecutive k of one row
    x = np.arange(32 * 64, dtype=np.float32).reshape(32, 64)
    f = frag_perm(x).reshape(-1, 8)
    assert all((np.diff(r) == 1).all() and int(r[0]) % 8 == 0 for r in f)
    assert np.array_equal(frag_unperm(frag_perm(x)), x)


def test_int8_fragment_major_decoder_linears():
    """int8 decoder weights permuted for v_mfma_i32_16x16x64_i8 — element (n, k) at
    ((n/16 * K/64 + k/64) * 64 + 16*((k/16)%4) + n%16) * 16 + k%16 — scales stay per row"""
    from oracle.whisper import OracleWhisper
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=6)
    h1, t1 = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_INT8_FLOAT16))
    assert h1.reserved == 6
    o = OracleWhisper(cfg, w, int8=True)
    for name in ("dec.0.self.qkv", "dec.1.ffn2", "dec.0.cross.out"):
        wq = o.q[name + ".w"][0].numpy().astype(np.int8)
        assert np.array_equal(t1[name + ".wq"], frag_perm(wq, 64)), name
        assert np.array_equal(t1[name + ".ws"], o.q[name + ".w"][1].numpy()), name
    # encoder and cross-K/V weights stay row-major
    for name in ("enc.0.ffn1", "dec.0.cross.kv"):
        assert np.array_equal(t1[name + ".wq"], o.q[name + ".w"][0].numpy().astype(np.int8)), name
    wq, ws = o.q["dec.tok_emb"]
    V = cfg.n_vocab
    un = frag_unperm(t1["dec.logits.wq"], 64)
    assert np.array_equal(un[:V], wq.numpy().astype(np.int8)) and not un[V:].any()
    assert np.array_equal(t1["dec.logits.ws"], ws.numpy())
