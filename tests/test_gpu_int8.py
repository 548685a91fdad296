"""compute_type="int8_float16" (kernel K25): the int8 MFMA GEMMs, the row
quantiser and the end-to-end engine against the oracle's int8 restatement (oracle/whisper.py, int8=True).

Integer accumulation is exact on both sides, so a single linear agrees to fp16 output rounding.  End to end
the only noise source is a LayerNorm / attention value that lands 1 fp16 ulp apart and flips one int8 code;
tolerances are those of the fp16 tests, loosened where a flipped code is visible."""
import numpy as np
import pytest

from conftest import bench_audio, check_hypothesis, make_model

pytestmark = pytest.mark.gpu

# Greedy ids are compared where the oracle's top-1 / top-2 margin exceeds MARGIN; beam hypotheses where every
# pruning-boundary gap exceeds it.  Measured on tiny.en (profiles/diag_int8.py): the engine's and the oracle's int8
# log-probs of a 2-token sequence differ by up to 3.5e-2, the difference between two candidates by 4.7e-2.  Both
# sides quantise fp16 values with scale = absmax / 127; when the two absmax elements of a row are one fp16 ulp apart
# (accumulation order), the scales differ by 1e-3 and every element within 0.13 * |x| / absmax of a rounding boundary
# — about 2 % of the row, up to 5 % — lands on the neighbouring code.
MARGIN = 8e-2


def _h(x):
    return x.astype(np.float16).astype(np.float32)


def _quant_rows(x):
    amax = np.abs(x).max(axis=-1).astype(np.float32)
    inv = np.where(amax > 0, np.float32(127.0) / np.where(amax > 0, amax, 1), 0).astype(np.float32)
    ds = np.where(amax > 0, amax / np.float32(127.0), 1).astype(np.float32)
    q = np.rint(x.astype(np.float32) * inv[..., None]).astype(np.int64)
    return q, ds


@pytest.fixture(scope="module")
def kmodel():
    cfg, w, m = make_model("micro", max_batch=2, max_beam=2, compute_type="int8_float16")
    return m


def _gemm_i8(model, A, W, bias=None, res=None, act=0):
    from faster_whisper_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((N, M) if act >= 2 else (M, N), dtype=np.float32)
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    bp = _lib.ptr(np.ascontiguousarray(bias, np.float32)) if bias is not None else None
    rp = _lib.ptr(np.ascontiguousarray(res, np.float32)) if res is not None else None
    _lib.check(lib.fw_test_gemm(model._replicas[0].handle, _lib.ptr(A), _lib.ptr(W), bp, rp, M, N, K, act, 1,
                                _lib.ptr(out)))
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (256, 384, 384), (300, 256, 128), (1500, 512, 512),
                                   (77, 128, 256), (1500, 128, 512), (1100, 256, 256)])
def test_gemm_int8_exact(kmodel, M, N, K):
    rng = np.random.default_rng(M + 3 * N + K)
    A = _h(rng.standard_normal((M, K)).astype(np.float32))
    A[min(5, M - 1)] = 0.0                       # an all-zero row: scale 1, codes 0
    W = _h((rng.standard_normal((N, K)) * (0.2 + np.arange(N)[:, None] / N)).astype(np.float32))
    b = _h(rng.standard_normal(N).astype(np.float32))
    out = _gemm_i8(kmodel, A, W, bias=b)
    aq, a_s = _quant_rows(A)
    wq, w_s = _quant_rows(W)
    acc = (aq @ wq.T).astype(np.float32)
    ref = _h(acc * a_s[:, None] * w_s[None, :] + b)
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"int8 gemm {M}x{N}x{K}: rel err vs integer reference {err:.2e}")
    assert err < 1e-3                             # fp16 output rounding only
    # and the quantised product is a sane approximation of the fp32 one
    full = A @ W.T + b
    assert np.abs(out - full).max() / np.abs(full).max() < 5e-2


@pytest.mark.parametrize("M,N,K", [(200, 128, 256), (1100, 256, 256)])   # one and many M tiles
def test_gemm_int8_transposed_epilogue(kmodel, M, N, K):
    rng = np.random.default_rng(9)
    A = _h(rng.standard_normal((M, K)).astype(np.float32))
    W = _h(rng.standard_normal((N, K)).astype(np.float32) * 0.3)
    b = _h(rng.standard_normal(N).astype(np.float32))
    out = _gemm_i8(kmodel, A, W, bias=b, act=2)   # act 2 = transposed store, no activation
    aq, a_s = _quant_rows(A)
    wq, w_s = _quant_rows(W)
    ref = _h((aq @ wq.T).astype(np.float32) * a_s[:, None] * w_s[None, :] + b).T
    assert out.shape == (N, M)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-3


@pytest.mark.parametrize("R,N,K", [(80, 1280, 1280), (80, 5120, 1280), (80, 1280, 5120), (77, 3840, 1280),
                                   (5, 128, 128), (333, 256, 512), (640, 1280, 1280),
                                   # merged runs: 4 x 4 tiles per workgroup from 1 024 rows on (ragged last row group)
                                   (1029, 1280, 1280), (1100, 1280, 5120)])
def test_dec_linear_int8_exact(kmodel, R, N, K):
    """the int8 decoder linear of a decode step (row quantiser writing fragment-major + v_mfma_i32_16x16x64_i8
    register-streaming GEMM) against the exact integer reference, at the large-v3 step shapes"""
    from faster_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(R + N + K)
    x = _h(rng.standard_normal((R, K)).astype(np.float32))
    x[min(3, R - 1)] = 0.0
    W = _h((rng.standard_normal((N, K)) * (0.2 + np.arange(N)[:, None] / N)).astype(np.float32))
    b = _h(rng.standard_normal(N).astype(np.float32))
    r = _h(rng.standard_normal((R, N)).astype(np.float32))
    out = np.empty((R, N), np.float32)
    out2 = np.empty((R, N), np.float32)
    _lib.check(lib.fw_test_dec_linear(kmodel._replicas[0].handle, _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), None, None,
                                      _lib.ptr(r), R, N, K, 0, 1, _lib.ptr(out), _lib.ptr(out2)))
    xq, x_s = _quant_rows(x)
    wq, w_s = _quant_rows(W)
    ref = _h((xq @ wq.T).astype(np.float32) * x_s[:, None] * w_s[None, :] + b + r)
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print(f"int8 dec linear {R}x{N}x{K}: rel err vs integer reference {err:.2e}")
    assert err < 1e-3


@pytest.fixture(scope="module", params=["micro", "tiny.en"])
def setup(request):
    from oracle.whisper import OracleWhisper
    cfg, w, model = make_model(request.param, seed=11, max_batch=4, max_beam=5, compute_type="int8_float16")
    oracle = OracleWhisper(cfg, w, int8=True)
    model._test_weights = w
    chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
    feats = model.log_mel(chunks)
    return cfg, model, oracle, feats


def _prompt(cfg, timestamps=False):
    p = list(cfg.sot_sequence)
    if not timestamps:
        p.append(cfg.no_timestamps)
    return p


def _suppress(cfg):
    return sorted({cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe, 1, 2, 7})


@pytest.mark.parametrize("R", [3, 15, 80, 83])
def test_dec_logits_int8(setup, R):
    """the int8 vocabulary projection of a decode step on the model's own (quantised) embedding: final LayerNorm ->
    per-row int8 -> v_mfma_i32_16x16x64_i8 over the whole vocabulary -> float32 logits, against the integer
    reference column by column (a wrong scale or a misplaced column tile shows as an O(1) error in its columns; a
    LayerNorm value that lands on the other side of a rounding boundary moves a logit by < 1 % of the row's range)"""
    from faster_whisper_amd import _lib
    cfg, model, oracle, _ = setup
    w = model._test_weights
    rng = np.random.default_rng(100 + R)
    x = _h((rng.standard_normal((R, cfg.d_model)) * 2 + 0.5).astype(np.float32))
    out = np.empty((R, cfg.n_vocab), np.float32)
    _lib.check(model._lib.fw_test_dec_logits(model._replicas[0].handle, _lib.ptr(x), R, _lib.ptr(out)))
    g, b = _h(w["dec.ln.g"].astype(np.float32)), _h(w["dec.ln.b"].astype(np.float32))
    mu = x.mean(-1, keepdims=True)
    xn = (x - mu) / np.sqrt(((x - mu) ** 2).mean(-1, keepdims=True) + 1e-5) * g + b
    xq, x_s = _quant_rows(xn.astype(np.float32))
    wq, w_s = _quant_rows(_h(w["dec.tok_emb"].astype(np.float32)))
    ref = (xq @ wq.T).astype(np.float32) * x_s[:, None] * w_s[None, :]
    scale = np.abs(ref).max()
    col_err = np.abs(out - ref).max(axis=0) / scale
    print(f"[{cfg.name}] int8 logits R={R}: worst column {int(col_err.argmax())} rel err {col_err.max():.2e}")
    assert np.isfinite(out).all() and col_err.max() < 1e-2


def test_compute_type_reported(setup):
    cfg, model, oracle, feats = setup
    assert model.compute_type == "int8_float16"


def test_encode_int8(setup):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    got = model.encode(StorageView.from_array(feats)).to_numpy()
    ref = oracle.encode(feats)
    err = float(np.abs(got - ref).max())
    rel = err / float(np.abs(ref).max())
    rms = float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    print(f"[{cfg.name}] int8 encoder: max abs err {err:.3e} (rel {rel:.2e}), rms rel {rms:.2e}")
    assert rel < 3e-2 and rms < 1.5e-2   # a flipped int8 code is ~1/127 of a row's absmax


def test_generate_int8_teacher_forced(setup):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg)
    kw = dict(beam_size=1, max_length=len(prompt) + 16, suppress_tokens=_suppress(cfg), length_penalty=0.0)
    got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, force_tokens=[g.sequences_ids[0] for g in got], **kw)
    for g, r in zip(got, ref):
        assert r.sequences_ids[0] == g.sequences_ids[0]
        print(f"[{cfg.name}] int8 teacher-forced cum logprob {g.scores[0]:.5f} vs {r.scores[0]:.5f}")
        assert abs(g.scores[0] - r.scores[0]) < 5e-3 * max(1.0, abs(r.scores[0]))


@pytest.mark.parametrize("beam", [1, 5])
def test_generate_int8(setup, beam):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg, timestamps=True)
    kw = dict(beam_size=beam, max_length=len(prompt) + 16, suppress_blank=True, suppress_tokens=_suppress(cfg),
              max_initial_timestamp_index=50)
    got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, **kw)
    for b, (g, r) in enumerate(zip(got, ref)):
        if beam == 1:
            n = 0
            while n < len(r.margins) and n < len(r.sequences_ids[0]) and r.margins[n] > MARGIN:
                n += 1
            assert g.sequences_ids[0][:n] == r.sequences_ids[0][:n]
        # score of the engine's own ids under the oracle always within 5e-3 (a flipped int8 code is visible);
        # different ids only when the two hypotheses are tied within 4e-2 under the oracle's scoring
        check_hypothesis(oracle, enc_np[b], prompt, g, r, kw, tol=5e-3, gap=4e-2, search=beam > 1, boundary=MARGIN,
                         what=f"[{cfg.name}] int8 beam={beam} chunk {b}")
        assert abs(g.no_speech_prob - r.no_speech_prob) < 2e-3


def test_int8_tracks_float16(setup):
    """the quantised engine stays close to the fp16 engine on the same weights (sanity of the whole int8 flow)"""
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    _, _, ref_model = make_model(cfg.name, seed=11, max_batch=4, max_beam=5)
    a = model.encode(StorageView.from_array(feats)).to_numpy()
    b = ref_model.encode(StorageView.from_array(feats)).to_numpy()
    rms = float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    print(f"[{cfg.name}] int8 vs fp16 encoder output: rms rel {rms:.3e}")
    assert rms < 0.1
