"""External anchor for the decoding rules of the (unpinned) CTranslate2 restatement: the timestamp / suppress
rules of oracle/whisper.py::_process_logits against the independent public implementation in the installed
`transformers` (generation/logits_process.py: WhisperTimeStampLogitsProcessor, SuppressTokensLogitsProcessor,
SuppressTokensAtBeginLogitsProcessor — themselves ports of openai-whisper's ApplyTimestampRules / SuppressTokens
/ SuppressBlank, which CTranslate2 ports too).  Random logits and random token histories: the set of forbidden
tokens and the resulting log-probabilities must be identical.  No GPU."""
import numpy as np
import pytest
import torch

from faster_whisper_amd import get_config, synthetic_weights
from oracle.whisper import OracleWhisper

lp = pytest.importorskip("transformers.generation.logits_process")


class _GenCfg:
    def __init__(self, cfg, mits):
        self.eos_token_id = cfg.eot
        self.bos_token_id = cfg.eot
        self.no_timestamps_token_id = cfg.no_timestamps
        self.max_initial_timestamp_index = mits
        self._detect_timestamp_from_logprob = True


def _history(rng, cfg, n):
    """a plausible decode history: text runs separated by timestamp pairs, non-decreasing times"""
    tb, out, t = cfg.timestamp_begin, [], int(rng.integers(0, 40))
    if n and rng.random() < 0.8:
        out.append(tb + t)
    while len(out) < n:
        r = rng.random()
        if r < 0.7:
            out.append(int(rng.integers(0, cfg.eot)))
        else:
            t += int(rng.integers(0, 60))
            out.append(tb + min(t, 1500))
            if rng.random() < 0.6 and len(out) < n:
                out.append(tb + min(t, 1500))
    return out[:n]


@pytest.fixture(scope="module")
def oracle():
    cfg = get_config("micro")
    return cfg, OracleWhisper(cfg, synthetic_weights(cfg, seed=2))


@pytest.mark.parametrize("mits", [50, 0, None])
def test_timestamp_rules_match_transformers(oracle, mits):
    cfg, o = oracle
    rng = np.random.default_rng(11 if mits is None else mits)
    P = 4
    prompt = list(cfg.sot_sequence) + [cfg.sot_prev]          # any P tokens: the processor only looks past begin_index
    proc = lp.WhisperTimeStampLogitsProcessor(_GenCfg(cfg, mits), begin_index=P)
    checked_e = 0
    for trial in range(400):
        n = int(rng.integers(0, 12))
        hist = _history(rng, cfg, n)
        logits = (rng.standard_normal(cfg.n_vocab) * 3).astype(np.float32)
        if trial % 3 == 0:          # make rule (e) fire sometimes: timestamp mass above the best text token
            logits[cfg.timestamp_begin:cfg.timestamp_begin + 200] += 4.0
        ids = torch.tensor([prompt + hist], dtype=torch.long)
        hf = proc(ids, torch.from_numpy(logits.copy())[None])[0]
        hf_lp = torch.log_softmax(hf.float(), dim=-1).numpy()
        mine = o._process_logits(logits, hist, True, None, False, mits, 1.0, 0, 0)
        dead_hf, dead_me = np.isneginf(hf_lp), np.isneginf(mine)
        assert np.array_equal(dead_hf, dead_me), (trial, hist, np.flatnonzero(dead_hf != dead_me)[:10])
        live = ~dead_me
        assert live.any() and np.abs(mine[live] - hf_lp[live]).max() < 2e-5
        checked_e += int(dead_me[:cfg.timestamp_begin].all() and n > 0 and not (hist[-1] >= cfg.timestamp_begin))
    assert checked_e > 5            # the "timestamp mass beats every text token" rule was exercised


def test_suppress_rules_match_transformers(oracle):
    cfg, o = oracle
    rng = np.random.default_rng(3)
    sup = sorted(set(int(x) for x in rng.integers(0, cfg.n_vocab, size=40)) | {cfg.sot, cfg.no_speech})
    begin = list(cfg.suppress_begin)
    p_sup = lp.SuppressTokensLogitsProcessor(sup)
    p_beg = lp.SuppressTokensAtBeginLogitsProcessor(begin, begin_index=3)
    mask = np.asarray(sup, dtype=np.int64)
    for n in (0, 1, 5):
        hist = [int(x) for x in rng.integers(0, cfg.eot, size=n)]
        logits = rng.standard_normal(cfg.n_vocab).astype(np.float32)
        ids = torch.tensor([[cfg.sot, cfg.lang_begin, cfg.transcribe] + hist])
        hf = p_beg(ids, p_sup(ids, torch.from_numpy(logits.copy())[None]))[0]
        hf_lp = torch.log_softmax(hf.float(), dim=-1).numpy()
        mine = o._process_logits(logits, hist, False, mask, True, 50, 1.0, 0, 0)
        assert np.array_equal(np.isneginf(hf_lp), np.isneginf(mine))
        live = ~np.isneginf(mine)
        assert np.abs(mine[live] - hf_lp[live]).max() < 2e-5
        assert np.isneginf(mine[begin]).all() == (n == 0)


def test_dtw_and_median_filter_match_transformers():
    """word-timestamp post-processing of `align` (median filter over time, DTW on the negated matrix) against
    transformers' ports of openai-whisper timing.py (generation_whisper.py: _median_filter, _dynamic_time_warping)"""
    gw = pytest.importorskip("transformers.models.whisper.generation_whisper")
    from oracle.whisper import _dtw, _median_filter
    rng = np.random.default_rng(8)
    for shape in [(1, 7, 40), (3, 12, 129), (2, 5, 9)]:
        x = rng.standard_normal(shape).astype(np.float32)
        for width in (7, 3):
            if shape[-1] <= width // 2:
                continue
            want = gw._median_filter(torch.from_numpy(x.copy()), width).numpy()
            got = _median_filter(x, width)
            assert got.shape == want.shape and np.array_equal(got, want)
    for n_tok, n_fr in [(6, 50), (13, 13), (20, 7), (1, 30), (9, 1)]:
        cost = rng.standard_normal((n_tok, n_fr))
        ti, fi = _dtw(cost.astype(np.float64))
        hti, hfi = gw._dynamic_time_warping(cost.astype(np.float64))
        assert np.array_equal(np.asarray(ti), hti) and np.array_equal(np.asarray(fi), hfi)


@pytest.mark.parametrize("penalty,ngram", [(1.3, 0), (1.0, 2), (1.0, 3), (1.15, 3)])
def test_repetition_penalty_and_ngram_rules_match_transformers(oracle, penalty, ngram):
    """CTranslate2's RepetitionPenalty / NoRepeatNgram as the oracle restates them (over the GENERATED tokens) against
    transformers' RepetitionPenaltyLogitsProcessor / NoRepeatNGramLogitsProcessor fed the same token history: the same
    penalised logits, the same banned tokens.  (Whether CTranslate2 also counts the prompt is [CT2-ext]; the reference's
    defaults — repetition_penalty 1, no_repeat_ngram_size 0, transcribe.py:752-753 — never reach that question.)"""
    cfg, o = oracle
    rng = np.random.default_rng(int(penalty * 100) + ngram)
    procs = []
    if penalty != 1.0:
        procs.append(lp.RepetitionPenaltyLogitsProcessor(penalty))
    if ngram:
        procs.append(lp.NoRepeatNGramLogitsProcessor(ngram))
    banned_any = 0
    for trial in range(200):
        n = int(rng.integers(1, 24))
        hist = [int(x) for x in rng.integers(0, 5, size=n)]        # a small alphabet: repeats and repeated n-grams are common
        logits = (rng.standard_normal(cfg.n_vocab) * 2).astype(np.float32)
        ids = torch.tensor([hist], dtype=torch.long)
        hf = torch.from_numpy(logits.copy())[None]
        for p in procs:
            hf = p(ids, hf)
        hf_lp = torch.log_softmax(hf[0].float(), dim=-1).numpy()
        mine = o._process_logits(logits, hist, False, None, False, 50, penalty, ngram, 0)
        assert np.array_equal(np.isneginf(hf_lp), np.isneginf(mine)), (trial, hist)
        live = ~np.isneginf(mine)
        assert np.abs(mine[live] - hf_lp[live]).max() < 2e-5
        banned_any += int(np.isneginf(mine).any())
    if ngram:
        assert banned_any > 20          # the n-gram rule fired
