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
    m._test_weights = w
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
                                   (1024, 256, 64), (1500, 768, 192), (1031, 256, 128),
                                   # long K (80 K tiles: the steady state of the DMA ring), one to three K tiles
                                   (700, 512, 5120), (260, 256, 192)])
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


@pytest.mark.parametrize("M,N,K", [(1504, 256, 128), (296, 128, 128), (1024, 512, 64), (2048, 200, 192)])
def test_gemm_transposed_output_staged_epilogue(model, M, N, K):
    """Round 5: the plain transposed epilogue (the encoder's V^T) goes through LDS and leaves as 256-byte row segments
    of Ct when the row stride allows 16-byte stores (here ldc = M, so M % 8 == 0 takes the staged form; the shapes of
    the test above have M % 8 != 0 and keep the direct stores).  Same arithmetic per element: the staged result equals
    the direct one (fw_test_knob 5 = 0) bit for bit, and the fp32 reference within fp16 round-off; N = 200 leaves the
    last column tile ragged (columns >= N are staged but never stored)."""
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + N)
    A = _h(rng.standard_normal((M, K)).astype(np.float32))
    W = _h((rng.standard_normal((N, K)) * (1.0 + np.arange(N)[:, None] / N)).astype(np.float32))
    b = _h(rng.standard_normal(N).astype(np.float32))
    out = {}
    try:
        for on in (0, 1):
            _lib.check(lib.fw_test_knob(5, on))
            out[on] = _gemm(model, A, W, bias=b, act=2)      # [N][M]
    finally:
        _lib.check(lib.fw_test_knob(5, 1))
    ref = (A @ W.T + b).T
    err = np.abs(out[1] - ref).max() / np.abs(ref).max()
    print(f"gemm transposed, staged epilogue {M}x{N}x{K}: rel err {err:.2e}; identical to the direct stores: "
          f"{np.array_equal(out[0], out[1])}")
    assert err < 2e-3
    assert np.array_equal(out[0], out[1])


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


@pytest.mark.parametrize("B,H,T,scale", [(1, 2, 64, 1.0), (2, 2, 200, 1.0), (1, 3, 1500, 1.0), (1, 2, 333, 3.0),
                                         (2, 1, 1500, 0.05)])
def test_encoder_attention(model, B, H, T, scale):
    """flash attention of the encoder against fp64.  scale 3: scores with a standard deviation of ~13 in the log2 domain —
    the softmax reference (kept on the matrix pipe, moved only when a query runs AE_LAG ahead of it) moves in most tiles, in
    both directions of key order; scale 0.05: it never moves after the first tile; the spike moves it late in the row"""
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(T)
    d = H * 64
    q = _h((scale * rng.standard_normal((B, T, d))).astype(np.float32))
    k = _h((scale * rng.standard_normal((B, T, d))).astype(np.float32))
    if scale > 1:      # keys sorted so that one head sees rising, the other falling scores along the key axis
        k[0, :, :64] = k[0, np.argsort(k[0, :, 0]), :64]
        k[0, :, 64:128] = k[0, np.argsort(-k[0, :, 64]), 64:128]
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
    # the kernel's one documented rounding point ahead of the MFMAs: Q' = fp16(q * log2(e) / 8), scores in the exp2 domain.
    # With scores of standard deviation 9 (scale 3) that rounding alone moves a near-tied pair of keys by 7e-3 of output
    # (numpy: fp64 with Q' against fp64 with q = 7.29e-3), so the large-score case is held against the reference WITH it
    qs = _h((q * (0.125 * 1.4426950408889634)).astype(np.float32)).reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    s2 = qs @ kh.transpose(0, 1, 3, 2)
    s2 -= s2.max(-1, keepdims=True)
    p2 = np.exp2(s2)
    p2 /= p2.sum(-1, keepdims=True)
    ref2 = (p2 @ vh).transpose(0, 2, 1, 3).reshape(B, T, d)
    err2 = np.abs(out - ref2).max()
    print(f"attention B={B} H={H} T={T} scale={scale}: abs err {err:.2e} (fp64), {err2:.2e} (fp64 with the kernel's fp16 Q')")
    assert np.isfinite(out).all() and err2 < 3e-3
    if scale <= 1:
        assert err < 5e-3


def test_encoder_attention_first_tile_far_below_zero(model):
    """the softmax reference is fixed from the FIRST 64-key tile, whatever its sign: when every score of that tile lies
    far below zero (here ~ -260 in the log2 domain, past the exp2 range of fp32) the reference must move down without
    rescaling the (still empty) accumulators by exp2(+260) = inf (0 x inf = NaN would poison the query's output and
    every later encoder layer)"""
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    B, H, T = 1, 2, 200
    d = H * 64
    q = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    k = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    v = _h(rng.standard_normal((B, T, d)).astype(np.float32))
    # head 0: every query has a large component along e0, the first 64 keys point the other way: q.k / 8 = -180 (-260 in
    # the log2 domain); head 1 the same with the LAST tile (the masked tail tile) far below instead
    q[0, :, 0] = 12.0
    k[0, :64, 0] = -120.0
    q[0, :, 64] = 12.0
    k[0, 192:, 64] = -120.0
    out = np.empty_like(q)
    _lib.check(lib.fw_test_attention(model._replicas[0].handle, _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), B, H, T,
                                     _lib.ptr(out)))
    qs = _h((q * (0.125 * 1.4426950408889634)).astype(np.float32)).reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    kh = k.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    vh = v.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    s2 = qs @ kh.transpose(0, 1, 3, 2)
    s2 -= s2.max(-1, keepdims=True)
    p2 = np.exp2(s2)
    p2 /= p2.sum(-1, keepdims=True)
    ref = (p2 @ vh).transpose(0, 2, 1, 3).reshape(B, T, d)
    assert np.isfinite(out).all(), "NaN / inf from a first key tile far below zero"
    err = np.abs(out - ref).max()
    print(f"attention, first tile ~260 below zero (log2 domain): abs err {err:.2e}")
    assert err < 3e-3


# ---- decoder-step linears: the fragment-major register-streaming GEMM exactly as a decode step launches it ----
def _dec_linear(model, x, W, bias=None, ln=None, res=None, act=0, int8=False):
    from faster_whisper_amd import _lib
    lib = _lib.load()
    R, K = x.shape
    N = W.shape[0]
    out = np.empty((R, N), np.float32)
    out_frag = np.empty((R, N), np.float32)
    f = lambda a: _lib.ptr(np.ascontiguousarray(a, np.float32)) if a is not None else None   # noqa: E731
    keep = [np.ascontiguousarray(a, np.float32) if a is not None else None
            for a in (x, W, bias, ln[0] if ln else None, ln[1] if ln else None, res)]
    ptrs = [_lib.ptr(a) if a is not None else None for a in keep]
    _lib.check(lib.fw_test_dec_linear(model._replicas[0].handle, *ptrs, R, N, K, act, int(int8), _lib.ptr(out),
                                      _lib.ptr(out_frag)))
    return out, out_frag


def _ln_ref(x, g, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-5) * g + b


# (R, N, K): the shapes of a large-v3 decode step at 16 chunks x beam 5 (80 rows: out-proj / cross-q, ffn1, ffn2 —
# K = 5120 runs the 8-wave instantiation and two load chunks per wave), a ragged row count with the fused-QKV
# width, a single row tile, and the row counts of merged decode runs (160 .. 640 rows: row groups on grid.y)
DEC_SHAPES = [(80, 1280, 1280), (80, 5120, 1280), (80, 1280, 5120), (77, 3840, 1280), (5, 128, 128), (16, 384, 1536),
              (160, 1280, 1280), (333, 256, 512), (640, 1280, 5120)]


@pytest.mark.parametrize("R,N,K", DEC_SHAPES)
def test_dec_linear_plain_bias_residual(model, R, N, K):
    rng = np.random.default_rng(R + N + K)
    x = _h(rng.standard_normal((R, K)).astype(np.float32))
    W = _h((rng.standard_normal((N, K)) * (0.5 + np.arange(N)[:, None] / N) / np.sqrt(K)).astype(np.float32))
    b = _h(0.1 * rng.standard_normal(N).astype(np.float32))
    r = _h(rng.standard_normal((R, N)).astype(np.float32))
    ref = x @ W.T + b + r
    out, out_frag = _dec_linear(model, x, W, bias=b, res=r)
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"dec linear {R}x{N}x{K} bias+res: rel err {err:.2e}")
    assert err < 2e-3
    assert np.array_equal(out, out_frag)       # the fragment-major copy is the same tensor


@pytest.mark.parametrize("R,N,K", [s for s in DEC_SHAPES if s[2] <= 1536])
def test_dec_linear_layernorm_folded_gelu(model, R, N, K):
    """qkv / cross-q / ffn1 form: LayerNorm folded into the weights (y = rstd*(W.g x - mu*s1) + cf), exact GELU"""
    rng = np.random.default_rng(R * 3 + N + K)
    x = _h((rng.standard_normal((R, K)) * 1.7 + 0.3).astype(np.float32))
    x[R // 2] *= 8.0                                   # one row with a very different scale
    x = _h(x)
    W = _h((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = _h(0.1 * rng.standard_normal(N).astype(np.float32))
    g = _h(1 + 0.1 * rng.standard_normal(K).astype(np.float32))
    lb = _h(0.05 * rng.standard_normal(K).astype(np.float32))
    for act in (0, 1):
        base = _ln_ref(x.astype(np.float64), g, lb) @ W.T.astype(np.float64) + b
        ref = _gelu(base) if act else base
        out, out_frag = _dec_linear(model, x, W, bias=b, ln=(g, lb), act=act)
        err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
        print(f"dec linear {R}x{N}x{K} LN-folded act={act}: rel err {err:.2e}")
        assert err < 4e-3
        assert np.array_equal(out, out_frag)


@pytest.mark.parametrize("M,N,K", [(1500, 1280, 256), (77, 128, 128), (1031, 256, 128), (1500, 384, 384), (32, 64, 64),
                                   (1473, 128, 192)])
@pytest.mark.parametrize("vt", [False, True])
def test_gemm_cross_kv_fragment_major(model, M, N, K, vt):
    """the cross-attention K / V^T projection epilogues (gemm.hip: the sub-tile staged through LDS, whole 1 KB runs of
    the MFMA-fragment-major layout dec_cross_attn_kernel streams): the hook un-permutes the device buffer on the host
    and fails if a padded key (>= M inside the last 32-key group, M = 1500 / 77 / 1031 / 1473: 4, 19, 25 and 31 of them)
    was written.  Asymmetric operands: a transposed or mis-grouped run cannot pass."""
    rng = np.random.default_rng(M + 3 * N + K + int(vt))
    A = _h((rng.standard_normal((M, K)) * (1.0 + np.arange(M)[:, None] / M)).astype(np.float32))
    W = _h((rng.standard_normal((N, K)) * (0.5 + np.arange(N)[:, None] / N)).astype(np.float32))
    b = _h(0.3 * rng.standard_normal(N).astype(np.float32))
    from faster_whisper_amd import _lib
    lib = _lib.load()
    out = np.zeros((M, N), np.float32)
    _lib.check(lib.fw_test_gemm(model._replicas[0].handle, _lib.ptr(np.ascontiguousarray(A)), _lib.ptr(np.ascontiguousarray(W)),
                                _lib.ptr(b), None, M, N, K, 9 if vt else 8, 0, _lib.ptr(out)))
    ref = A @ W.T + b
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"cross-{'V^T' if vt else 'K'} projection {M}x{N}x{K}: rel err {err:.2e}")
    assert err < 2e-3


BIG_CFGS = (0, 1, 2)     # workgroup shapes of dec_gemm_big_kernel (dec_kernels.hip: launch_dec_gemm_big)


@pytest.mark.parametrize("R", [5, 80, 333, 800, 960, 1521, 1680])
def test_dec_linear_big_bit_identical(model, R):
    """the GEMM-shaped decoder linear of merged runs (dec_gemm_big_kernel: 64 x 64 outputs per wave, fragments staged
    once per workgroup tile in LDS) must return EXACTLY the bits of the skinny kernel of solo runs at every large-v3
    step shape — it keeps that kernel's K slices, MFMA chains and addition order, and both end in the same pinned
    epilogue — so that a merged decode run stays bit-identical to a solo run whichever kernel serves it.  Every
    workgroup shape is checked (ragged row and column tile tails included: 333, 1521 rows)"""
    rng = np.random.default_rng(900 + R)
    for N, K, ln, act, use_res in ((1280, 1280, False, 0, True), (3840, 1280, True, 0, False),
                                   (5120, 1280, True, 1, False), (1280, 5120, False, 0, True)):
        x = _h((rng.standard_normal((R, K)) * 1.3 + 0.2).astype(np.float32))
        W = _h((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        b = _h(0.1 * rng.standard_normal(N).astype(np.float32))
        r = _h(rng.standard_normal((R, N)).astype(np.float32)) if use_res else None
        lnp = (_h(1 + 0.1 * rng.standard_normal(K).astype(np.float32)),
               _h(0.05 * rng.standard_normal(K).astype(np.float32))) if ln else None
        a, a_frag = _dec_linear(model, x, W, bias=b, ln=lnp, res=r, act=act, int8=5)    # the skinny kernel
        # and against fp64, so that "identical" cannot mean "identically wrong"
        base = (_ln_ref(x.astype(np.float64), *lnp) if ln else x.astype(np.float64)) @ W.T.astype(np.float64) + b
        ref = (_gelu(base) if act else base) + (r if use_res else 0.0)
        err = np.abs(a - ref).max() / max(1.0, np.abs(ref).max())
        assert err < 4e-3, (N, K, err)
        # 0: what a decode step launches at this row count; 6 / 7: one tile / 2 x 2 tiles per workgroup
        for variant in (0, 6, 7) + tuple(10 + c for c in BIG_CFGS):
            t, t_frag = _dec_linear(model, x, W, bias=b, ln=lnp, res=r, act=act, int8=variant)
            same = np.array_equal(a, t)
            print(f"big[{variant}] vs skinny {R}x{N}x{K} ln={ln} act={act}: identical={same}, "
                  f"max diff {np.abs(a - t).max():.2e}")
            assert same and np.array_equal(a_frag, t_frag), (variant, N, K)


def test_dec_linear_big_odd_shapes(model):
    """column counts that do not fill a workgroup tile, a single row tile, tiny.en's d = 384 (K = 384: 3 k-steps per slice)"""
    rng = np.random.default_rng(5)
    for R, N, K in ((17, 384, 384), (130, 1536, 384), (600, 384, 1536), (257, 1280, 1280)):
        x = _h(rng.standard_normal((R, K)).astype(np.float32))
        W = _h((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        b = _h(0.1 * rng.standard_normal(N).astype(np.float32))
        a, a_frag = _dec_linear(model, x, W, bias=b, int8=5)
        ref = x @ W.T + b
        assert np.abs(a - ref).max() / max(1.0, np.abs(ref).max()) < 2e-3
        n_ok = 0
        for c in BIG_CFGS:
            try:
                t, t_frag = _dec_linear(model, x, W, bias=b, int8=10 + c)
            except RuntimeError:        # a k-steps-per-stage that does not divide this K's slices: the launcher refuses
                continue
            n_ok += 1
            assert np.array_equal(a, t) and np.array_equal(a_frag, t_frag), (c, R, N, K)
        assert n_ok >= 2, (R, N, K)


@pytest.mark.parametrize("R", [1, 5, 16, 48, 64, 80, 83, 640])
def test_dec_logits_projection(model, R):
    """the vocabulary projection of a decode step (full-K-per-wave kernel, final LayerNorm folded) against fp64 on
    the model's own weights; R covers every row-tile instantiation and the row groups of merged runs"""
    from faster_whisper_amd import _lib
    lib = _lib.load()
    cfg = model.config
    w = model._test_weights
    rng = np.random.default_rng(R)
    x = _h((rng.standard_normal((R, cfg.d_model)) * 2 + 0.5).astype(np.float32))
    out = np.empty((R, cfg.n_vocab), np.float32)
    _lib.check(lib.fw_test_dec_logits(model._replicas[0].handle, _lib.ptr(x), R, _lib.ptr(out)))
    E = w["dec.tok_emb"].astype(np.float64)
    ref = _ln_ref(x.astype(np.float64), w["dec.ln.g"].astype(np.float64), w["dec.ln.b"].astype(np.float64)) @ E.T
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"dec logits R={R}: rel err {err:.2e}")
    assert err < 3e-3
