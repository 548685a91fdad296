"""The logits-rules kernel (dec_kernels.hip: dec_logits_process_kernel — suppress lists, repetition penalty, no-repeat
n-gram, the five timestamp rules, fp32 log-softmax, top-2K candidates of cum + logp; Gumbel arg-max when sampling)
against the oracle's restatement of CTranslate2's logits processors (oracle/whisper.py::_process_logits, SURVEY.md
A.3), ONE launch on seeded random logits through the C ABI test hook fw_test_logits_rules.

The comparison is id for id: every row holds DISTINCT multiples of 1/4096, so two candidates are never closer than
2.4e-4 and the order is decided exactly; candidate values agree to float32 round-off
of the log-sum-exp.  Rule (e) (timestamp mass vs best text token) is exercised on both sides of its threshold with a
margin far above that round-off."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["tiny.en", "large-v3"])
def env(request):
    """a model is needed only for its config (token ids, vocabulary): decoder-less geometry is irrelevant here, so the
    large-v3 vocabulary (51 866, multilingual ids) rides on a micro-sized network"""
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    from oracle.whisper import OracleWhisper
    import dataclasses
    full = get_config(request.param)
    micro = get_config("micro")
    keep = ("name", "n_mels", "d_model", "n_heads", "n_enc_layers", "n_dec_layers")
    cfg = dataclasses.replace(full, **{k: getattr(micro, k) for k in keep if hasattr(micro, k)})
    w = synthetic_weights(cfg, seed=5)
    model = Whisper("synthetic:rules", device="cuda", files={"config": cfg, "weights": w}, max_batch_size=2,
                    max_beam_size=5)
    return cfg, model, OracleWhisper(cfg, w)


def _logits(cfg, R, rng, scale=1.0):
    """every row: V DISTINCT multiples of 1/4096 in [0, 16) (times `scale`, a power of two), shuffled"""
    V = cfg.n_vocab
    assert V <= 65536
    base = rng.permutation(65536)[:V].astype(np.float32) / np.float32(4096.0)
    out = np.stack([rng.permutation(base) for _ in range(R)]) * np.float32(scale)
    return np.ascontiguousarray(out, dtype=np.float32)


def _launch(model, logits, hist, cum, K, with_ts, **opt):
    from faster_whisper_amd import _lib
    R, V = logits.shape
    n = len(hist[0]) if hist is not None and len(hist) else 0
    o = _lib.FwGenOpts()
    o.beam_size, o.patience, o.num_hypotheses, o.length_penalty = K, 1.0, 1, 1.0
    o.repetition_penalty = float(opt.get("repetition_penalty", 1.0))
    o.no_repeat_ngram_size = int(opt.get("no_repeat_ngram_size", 0))
    o.max_length = 448
    o.max_initial_timestamp_index = int(opt.get("max_initial_timestamp_index", 50))
    o.suppress_blank = int(opt.get("suppress_blank", True))
    sup = np.asarray(opt.get("suppress_tokens") or [], dtype=np.int32)
    o.suppress_tokens = _lib.as_i32p(sup) if sup.size else None
    o.n_suppress_tokens = int(sup.size)
    o.sampling_topk = int(opt.get("sampling_topk", 1))
    o.sampling_temperature = float(opt.get("sampling_temperature", 1.0))
    o.seed = int(opt.get("seed", 0))
    o.min_new_tokens = int(opt.get("min_new_tokens", 0))
    Cn = 1 if (K == 1 and o.sampling_topk != 1) else 2 * K
    h = np.ascontiguousarray(np.asarray(hist, dtype=np.int32).reshape(R, n)) if n else np.zeros((R, 1), np.int32)
    cv = np.zeros((R, Cn), np.float32)
    ct = np.zeros((R, Cn), np.int32)
    cum = np.ascontiguousarray(cum, dtype=np.float32)
    _lib.check(model._lib.fw_test_logits_rules(model._replicas[0].handle, _lib.ptr(logits), R, _lib.ptr(h), n,
                                               _lib.ptr(cum), C.byref(o), int(with_ts), _lib.ptr(cv), _lib.ptr(ct)))
    return cv, ct


def _reference(oracle, logits, hist, cum, K, with_ts, **opt):
    cfg = oracle.cfg
    V = logits.shape[1]
    mask = None
    if opt.get("suppress_tokens"):
        mask = np.zeros(V, dtype=bool)
        mask[[t for t in opt["suppress_tokens"] if 0 <= t < V]] = True
    vals, toks = [], []
    for r in range(logits.shape[0]):
        lp = oracle._process_logits(logits[r], list(hist[r]) if hist is not None else [], with_ts, mask,
                                    bool(opt.get("suppress_blank", True)), int(opt.get("max_initial_timestamp_index", 50)),
                                    float(opt.get("repetition_penalty", 1.0)), int(opt.get("no_repeat_ngram_size", 0)),
                                    int(opt.get("min_new_tokens", 0)))
        order = np.lexsort((np.arange(V), -lp))[:2 * K]          # value descending, index ascending
        v = np.float32(cum[r]) + lp[order]
        t = order.copy()
        dead = ~np.isfinite(lp[order])
        v[dead], t[dead] = -np.inf, 0
        vals.append(v)
        toks.append(t)
    return np.stack(vals).astype(np.float32), np.stack(toks).astype(np.int32)


def _compare(env, hist, K, with_ts, seed, scale=1.0, edit=None, **opt):
    cfg, model, oracle = env
    rng = np.random.default_rng(seed)
    R = 2 * K
    logits = _logits(cfg, R, rng, scale)
    if edit is not None:
        edit(logits)
    cum = -rng.random(R).astype(np.float32) * 3
    hist = None if hist is None else [list(hist) for _ in range(R)]
    cv, ct = _launch(model, logits, hist, cum, K, with_ts, **opt)
    rv, rt = _reference(oracle, logits, hist, cum, K, with_ts, **opt)
    assert np.array_equal(ct, rt), (ct[0], rt[0])
    fin = np.isfinite(rv)
    assert np.array_equal(np.isfinite(cv), fin)
    assert np.abs(cv[fin] - rv[fin]).max() < 2e-5
    return cv, ct


def _sup(cfg):
    return sorted({cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, 1, 2, 7, 63, 64, 1023, 1024, cfg.n_vocab - 1})


def test_plain_and_suppress_lists(env):
    cfg = env[0]
    _compare(env, None, 5, False, 1, suppress_blank=False)
    _compare(env, None, 5, False, 2, suppress_blank=True, suppress_tokens=_sup(cfg))
    _compare(env, [11, 12, 13], 5, False, 3, suppress_tokens=_sup(cfg))
    _compare(env, [11, 12], 1, False, 4, suppress_tokens=_sup(cfg), min_new_tokens=5)   # <eot> held back
    _compare(env, [11] * 447 if cfg.n_text_ctx > 447 else [11] * (cfg.n_text_ctx - 1), 2, False, 5)   # longest history


def test_repetition_penalty_and_no_repeat_ngram(env):
    cfg = env[0]
    h = [20, 21, 22, 20, 21, 23, 20, 21]
    _compare(env, h, 5, False, 6, repetition_penalty=1.3)
    _compare(env, h, 5, False, 7, no_repeat_ngram_size=3)          # 20 21 -> 22 and 23 are forbidden
    _compare(env, h, 3, False, 8, repetition_penalty=0.8, no_repeat_ngram_size=2, suppress_tokens=_sup(cfg))
    _compare(env, [5], 5, False, 9, no_repeat_ngram_size=1)


def test_timestamp_rules(env):
    cfg = env[0]
    tb = cfg.timestamp_begin
    sup = _sup(cfg)

    def quiet_timestamps(lg):
        lg[:, tb:] -= 8.0      # timestamp mass well below the best text token: rule (e) leaves the text ids alone

    # with the raw draw the 1 501 timestamp logits outweigh the best text token (rule (e) masks the text ids)
    for K in (1, 5):
        for edit in (quiet_timestamps, None):
            _compare(env, None, K, True, 10, edit=edit, suppress_tokens=sup)             # first token: a timestamp <= tb+50
            _compare(env, None, K, True, 11, edit=edit, max_initial_timestamp_index=0)
            _compare(env, None, K, True, 12, edit=edit, max_initial_timestamp_index=-1)
            _compare(env, [tb + 3], K, True, 13, edit=edit, suppress_tokens=sup)         # after the opening timestamp: text only
            _compare(env, [tb + 3, 40], K, True, 14, edit=edit)                          # text, or timestamps > tb+3
            _compare(env, [tb + 3, 40, tb + 9], K, True, 15, edit=edit)                  # open pair: timestamp >= tb+9 or <eot>
            _compare(env, [tb + 3, 40, tb + 9, tb + 9], K, True, 16, edit=edit)          # closed pair: text only
            _compare(env, [tb + 3, 40, tb + 9, tb + 9, 41, 42], K, True, 17, edit=edit, suppress_tokens=sup)   # > tb+9


def test_timestamp_mass_rule_both_sides(env):
    """rule (e): when logsumexp(timestamp logits) > max(text logit) the text ids are masked"""
    cfg = env[0]
    tb = cfg.timestamp_begin

    def favour_text(lg):
        lg[:, 100] = lg.max() + 8.0          # one text token far above the whole timestamp mass

    def favour_timestamps(lg):
        lg[:, tb + 20:tb + 60] += 8.0        # the timestamp mass far above the best text token

    for K in (1, 5):
        _, ct = _compare(env, [tb + 3, 40], K, True, 20, edit=favour_text)
        assert (ct[:, 0] == 100).all()
        _, ct = _compare(env, [tb + 3, 40], K, True, 21, edit=favour_timestamps)
        assert (ct >= tb).all()


def test_everything_masked(env):
    """closed timestamp pair (no timestamps) + every text id and <eot> suppressed: no candidate at all"""
    cfg = env[0]
    tb = cfg.timestamp_begin
    cv, ct = _launch(env[1], _logits(cfg, 2, np.random.default_rng(3)), [[tb + 1, tb + 1]] * 2, np.zeros(2, np.float32), 1,
                     True, suppress_tokens=list(range(tb)))
    assert np.isneginf(cv).all() and (ct == 0).all()


def test_gumbel_sampling_draw(env):
    """sampling: arg-max of logp / T + Gumbel noise (counter-based hash of seed, row, step, token), recorded score =
    cum + logp of the drawn token"""
    from oracle.whisper import _gumbel
    cfg, model, oracle = env
    rng = np.random.default_rng(30)
    R, V = 4, cfg.n_vocab
    logits = _logits(cfg, R, rng, scale=0.5)
    hist = [[11, 12, 13]] * R
    cum = -rng.random(R).astype(np.float32)
    for seed, T in ((12345, 1.0), ((7 << 32) | 99, 0.7)):
        cv, ct = _launch(model, logits, hist, cum, 1, False, sampling_topk=0, sampling_temperature=T, seed=seed,
                         suppress_tokens=_sup(cfg))
        mask = np.zeros(V, dtype=bool)
        mask[_sup(cfg)] = True
        for r in range(R):
            lp = oracle._process_logits(logits[r], hist[r], False, mask, True, 50, 1.0, 0, 0)
            key = np.where(np.isfinite(lp), lp * np.float32(1.0 / T) + _gumbel(seed, r, 3, V), -np.inf)
            top2 = np.sort(key)[-2:]
            assert top2[1] - top2[0] > 1e-4, "degenerate draw in the test data"
            assert ct[r, 0] == int(np.argmax(key))
            assert abs(cv[r, 0] - (cum[r] + lp[ct[r, 0]])) < 2e-5
