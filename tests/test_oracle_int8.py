"""int8_float16 restatement (oracle/whisper.py, int8=True) and the host-side int8 weight packer of the C-ABI
library (runs without a GPU): known-answer quantisation, exactness of the integer product, and bit-identity
of the packed int8 weights / scales / de-quantised embedding with the oracle's."""
import ctypes as C

import numpy as np
import torch

from faster_whisper_amd import _lib, get_config, synthetic_weights
from faster_whisper_amd.backend import pack_blob
from oracle.whisper import OracleWhisper


def test_quant_rows_known_answers():
    x = torch.tensor([[1.0, -2.0, 0.5, 0.0],
                      [0.0, 0.0, 0.0, 0.0],
                      [0.25, 0.75, -0.125, 127.0]])
    q, ds = OracleWhisper._quant_rows(x)
    # row 0: 127/2 = 63.5 -> 63.5 (tie -> even 64), -127, 31.75 -> 32, 0
    assert q[0].tolist() == [64, -127, 32, 0]
    assert float(ds[0]) == np.float32(2.0) / np.float32(127.0)
    # all-zero row: codes 0, factor 1
    assert q[1].tolist() == [0, 0, 0, 0] and float(ds[1]) == 1.0
    # row 2: scale 1 -> plain round-half-even
    assert q[2].tolist() == [0, 1, 0, 127]
    assert int(q.abs().max()) <= 127


def test_qmatmul_is_exact_integer_product():
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=3)
    o = OracleWhisper(cfg, w, int8=True)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((7, cfg.d_model)).astype(np.float32)).half().float()
    key = "dec.0.ffn1.w"
    y = o._qmatmul(x, key)
    wq, ws = o.q[key]
    xq, xs = o._quant_rows(x)
    acc = (xq.numpy().astype(np.int64) @ wq.numpy().astype(np.int64).T)
    ref = acc.astype(np.float32) * xs.numpy()[:, None] * ws.numpy()[None, :]
    assert np.array_equal(y.numpy(), ref)
    # quantisation error against the float product stays at the percent level
    full = (x @ o.w[key].t()).numpy()
    assert np.abs(y.numpy() - full).max() / np.abs(full).max() < 3e-2


def test_qmatmul_is_exact_at_the_largest_magnitudes():
    """every code +-127 over K = 5120 (the largest reduction of large-v3): |acc| reaches 127 * 127 * 5120 = 8.3e7, past
    what ONE float32 accumulation holds exactly — the blocked product must still give the int64 integers"""
    cfg = get_config("micro")
    o = OracleWhisper(cfg, synthetic_weights(cfg, seed=3), int8=True)
    rng = np.random.default_rng(4)
    K, N = 5120, 96
    sw = rng.choice([-1.0, 1.0], size=(N, K)).astype(np.float32)
    sw[0], sw[1] = 1.0, -1.0                          # a row that adds up to +-127 * 127 * K against an all-ones x
    sx = rng.choice([-1.0, 1.0], size=(5, K)).astype(np.float32)
    sx[0] = 1.0
    o.q["worst"] = OracleWhisper._quant_rows(torch.from_numpy(sw * 0.5))
    x = torch.from_numpy(sx * 3.0)
    y = o._qmatmul(x, "worst")
    wq, ws = o.q["worst"]
    xq, xs = o._quant_rows(x)
    assert float(wq.abs().min()) == 127.0 and float(xq.abs().min()) == 127.0
    acc = xq.numpy().astype(np.int64) @ wq.numpy().astype(np.int64).T
    assert int(np.abs(acc).max()) == 127 * 127 * K
    assert np.array_equal(y.numpy(), acc.astype(np.float32) * xs.numpy()[:, None] * ws.numpy()[None, :])


def test_int8_oracle_tracks_float_oracle():
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=5)
    a = OracleWhisper(cfg, w, emulate_fp16=True)
    b = OracleWhisper(cfg, w, int8=True)
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((1, cfg.n_mels, 3000)).astype(np.float32) * 0.3
    ea, eb = a.encode(feats), b.encode(feats)
    rms = float(np.sqrt(np.mean((ea - eb) ** 2)) / np.sqrt(np.mean(ea ** 2)))
    assert 0 < rms < 0.1
    # embedding lookup returns the de-quantised shared projection weight
    wq, ws = b.q["dec.tok_emb"]
    assert np.array_equal(b.w["dec.tok_emb"].numpy(),
                          (wq.float() * ws[:, None]).half().float().numpy())


class _Hdr(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_int32), ("n_tensors", C.c_int32),
                ("total_bytes", C.c_int64), ("compute_type", C.c_int32), ("reserved", C.c_int32),
                ("cfg", _lib.FwConfig)]


class _Entry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("dtype", C.c_int32), ("ndim", C.c_int32), ("dims", C.c_int64 * 4),
                ("offset", C.c_int64), ("nbytes", C.c_int64)]


def _parse_blob(blob: np.ndarray):
    raw = blob.tobytes()
    h = _Hdr.from_buffer_copy(raw[:C.sizeof(_Hdr)])
    assert h.magic == b"FWAMDBL1"
    out = {}
    pos = C.sizeof(_Hdr)
    for _ in range(h.n_tensors):
        e = _Entry.from_buffer_copy(raw[pos:pos + C.sizeof(_Entry)])
        pos += C.sizeof(_Entry)
        dt = {0: np.float32, 1: np.float16, 2: np.int8}[e.dtype]
        shape = tuple(e.dims[i] for i in range(e.ndim))
        out[e.name.decode()] = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=e.offset).reshape(shape)
    return h, out


def test_int8_blob_matches_oracle_quantisation():
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=9)
    blob = pack_blob(cfg, w, _lib.COMPUTE_INT8_FLOAT16)
    h, t = _parse_blob(blob)
    assert h.compute_type == _lib.COMPUTE_INT8_FLOAT16
    o = OracleWhisper(cfg, w, int8=True)
    checked = 0
    from conftest import frag_unperm
    for name in ["enc.0.attn.qkv", "enc.1.ffn2", "dec.0.self.qkv", "dec.1.cross.kv", "dec.0.ffn1", "dec.1.cross.out"]:
        wq, ws = o.q[name + ".w"]
        stored = t[name + ".wq"]
        if name.startswith("dec.") and "cross.kv" not in name:
            stored = frag_unperm(stored, 64)      # the per-layer decoder linears are stored MFMA-fragment-major
        assert np.array_equal(stored.astype(np.int32), wq.numpy()), name
        assert np.array_equal(t[name + ".ws"], ws.numpy()), name
        assert t[name + ".b"].dtype == np.float16
        checked += 1
    assert checked == 6
    wq, ws = o.q["dec.tok_emb"]
    assert np.array_equal(frag_unperm(t["dec.logits.wq"], 64)[:cfg.n_vocab].astype(np.int32), wq.numpy())
    assert np.array_equal(t["dec.logits.ws"], ws.numpy())
    assert np.array_equal(t["dec.tok_emb"].astype(np.float32), o.w["dec.tok_emb"].numpy())
    # LayerNorms stay explicit (not folded) in int8 mode; convolutions stay fp16
    assert "dec.0.ln1.g" in t and "dec.0.self.qkv.wf" not in t
    assert t["enc.conv1.wg"].dtype == np.float16
    # the fp16 blob of the same weights has the folded form instead
    h16, t16 = _parse_blob(pack_blob(cfg, w, _lib.COMPUTE_FLOAT16))
    assert "dec.0.self.qkv.wf" in t16 and "dec.0.self.qkv.wq" not in t16
    assert blob.nbytes < 0.75 * pack_blob(cfg, w, _lib.COMPUTE_FLOAT16).nbytes
