"""Single-kernel parity (through the C-ABI test hooks) against plain float32 references."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import make_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    cfg, w, m = make_model("micro", max_batch=2, max_beam=2)
    return m


def _h(x):  # fp16 rounding (what the engine stores)
    return x.astype(np.float16).astype(np.float32)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def _gemm(model, A, W, bias=None, res=None, act=0):
    from faster_whisper_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((N, M) if act >= 2 else (M, N), dtype=np.float32)
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    bp = _lib.ptr(np.ascontiguousarray(bias, np.float32)) if bias is not None else None
    rp = _lib.ptr(np.ascontiguousarray(res, np.float32)) if res is not None else None
    _lib.check(lib.fw_test_gemm(model._replicas[0].handle, _lib.ptr(A), _lib.ptr(W), bp, rp, M, N, K, act, 0,
                                _lib.ptr(out)))
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 384), (300, 256, 128), (1500, 512, 1280),
                                   (3000, 128, 384), (77, 128, 256),
                                   # many M tiles, ragged last tile
                                   (1024, 256, 64), (1500, 768, 192), (1031, 256, 128)])
def test_gemm_plain(model, M, N, K):
    rng = np.random.default_rng(M * 7 + N + K)
    A = _h(rng.standard_normal((M, K)).astype(np.float32))
    # asymmetric W so a transposed C write cannot pass
    W = _h((rng.standard_normal((N, K)) * (1.0 + np.arange(N)[:, None] / N)).astype(np.float32))
    out = _gemm(model, A, W)
    ref = A @ W.T
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"gemm {M}x{N}x{K}: rel err {err:.2e}")
    assert err < 2e-3


@pytest.mark.parametrize("M,N,K", [(200, 256, 192), (1100, 256, 192)])   # one and many M tiles
def test_gemm_epilogues(model, M, N, K):
    rng = np.random.default_rng(5)
    A = _h(rng.standard_normal((M, K)).astype(np.float32) * 0.5)
    W = _h(rng.standard_normal((N, K)).astype(np.float32) * 0.2)
    b = _h(rng.standard_normal(N).astype(np.float32))
    r = _h(rng.standard_normal((M, N)).astype(np.float32))
    base = A @ W.T + b
    for act, res, ref in [(0, None, base), (1, None, _gelu(base)), (0, r, base + r), (1, r, _gelu(base) + r)]:
        out = _gemm(model, A, W, bias=b, res=res, act=act)
        err = np.abs(out - ref).max()
        print(f"gemm epilogue act={act} res={res is not None}: abs err {err:.2e}")
        assert err < 2e-2 * max(1.0, np.abs(ref).max() / 4)


@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (1500, 256, 128), (1027, 512, 64)])
def test_gemm_transposed_output(model, M, N, K):
    rng = np.random.default_rng(6)
    A = _h(rng.standard_normal((M, K)).astype(np.float32))
    W = _h(rng.standard_normal((N, K)).astype(np.float32))
    b = _h(rng.standard_normal(N).astype(np.float32))
    out = _gemm(model, A, W, bias=b, act=2)  # [N][M]
    ref = (A @ W.T + b).T
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print(f"gemm transposed: rel err {err:.2e}")
    assert err < 2e-3


@pytest.mark.parametrize("d", [128, 384, 1280])
def test_layernorm(model, d):
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(d)
    rows = 37
    x = _h((rng.standard_normal((rows, d)) * 3 + 1).astype(np.float32))
    g = _h(1 + 0.1 * rng.standard_normal(d).astype(np.float32))
    b = _h(0.1 * rng.standard_normal(d).astype(np.float32))
    out = np.empty_like(x)
    _lib.check(lib.fw_test_layernorm(model._replicas[0].handle, _lib.ptr(x), _lib.ptr(g), _lib.ptr(b), rows, d,
                                     _lib.ptr(out)))
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    ref = (x - mu) / np.sqrt(var + 1e-5) * g + b
    err = np.abs(out - ref).max()
    print(f"layernorm d={d}: abs err {err:.2e}")
    assert err < 4e-3


@pytest.mark.parametrize("B,H,T", [(1, 2, 64), (2, 2, 200), (1, 3, 1500)])
def test_encoder_attention(model, B, H, T):
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(T)
    d = H * 64
    q = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    k = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    v = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    # spike one key against one query so the online-softmax rescale branch is exercised late in the row
    k[0, T - 3, :64] = q[0, 5, :64] * 4
    k = _h(k)
    out = np.empty_like(q)
    _lib.check(lib.fw_test_attention(model._replicas[0].handle, _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), B, H, T,
                                     _lib.ptr(out)))
    qh = q.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    kh = k.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    vh = v.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    s = qh @ kh.transpose(0, 1, 3, 2) / 8.0
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ vh).transpose(0, 2, 1, 3).reshape(B, T, d)
    err = np.abs(out - ref).max()
    print(f"attention B={B} H={H} T={T}: abs err {err:.2e}")
    assert err < 5e-3
