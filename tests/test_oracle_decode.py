"""Known-answer tests of the decoding rules restated in oracle/whisper.py:
suppress lists, suppress-blank, repetition penalty, no-repeat-ngram, the five timestamp rules,
score normalisation, greedy vs beam bookkeeping on scripted logits."""
import numpy as np
import pytest
import torch

from faster_whisper_amd import get_config
from oracle.whisper import OracleWhisper, _dtw, _median_filter, _topk_stable, max_new_tokens


class _Scripted(OracleWhisper):
    """oracle with the network replaced by a table: logits depend only on (last token, position)"""

    def __init__(self, cfg, table_fn):
        self.cfg = cfg
        self.h = False
        self.d, self.H = cfg.d_model, cfg.n_heads
        self.table_fn = table_fn
        self.w = {}

    def cross_kv(self, enc):
        return [(torch.zeros(1, 1, 1, 1), torch.zeros(1, 1, 1, 1))] * self.cfg.n_dec_layers

    def decoder_step(self, tok, pos, cache, ckv):
        cache.k = [torch.zeros(tok.shape[0], 1, 1, 1)] * self.cfg.n_dec_layers
        cache.v = cache.k
        return torch.stack([torch.tensor([float(t), float(pos)]) for t in tok.tolist()])

    def logits(self, hidden):
        return torch.stack([torch.from_numpy(self.table_fn(int(h[0]), int(h[1])).astype(np.float32)) for h in hidden])


@pytest.fixture(scope="module")
def cfg():
    return get_config("micro")


def _proc(o, lg, gen, **kw):
    args = dict(with_timestamps=False, suppress_mask=None, suppress_blank=False, max_initial_timestamp_index=50,
                repetition_penalty=1.0, no_repeat_ngram_size=0, min_new_tokens=0)
    args.update(kw)
    return o._process_logits(lg, gen, args["with_timestamps"], args["suppress_mask"], args["suppress_blank"],
                             args["max_initial_timestamp_index"], args["repetition_penalty"],
                             args["no_repeat_ngram_size"], args["min_new_tokens"])


def test_suppress_and_blank_and_logsoftmax(cfg):
    o = _Scripted(cfg, None)
    lg = np.zeros(cfg.n_vocab, np.float32)
    lp = _proc(o, lg, [])
    assert np.allclose(lp, -np.log(cfg.n_vocab), atol=1e-5)
    lp = _proc(o, lg, [], suppress_mask=np.array([3, 9]), suppress_blank=True)
    assert np.isneginf(lp[[3, 9, 5, cfg.eot]]).all()          # 5 = " " token of the micro vocabulary
    assert np.isclose(np.exp(lp[np.isfinite(lp)]).sum(), 1.0, atol=1e-4)
    lp = _proc(o, lg, [17], suppress_blank=True)               # only at the first generated step
    assert np.isfinite(lp[5]) and np.isfinite(lp[cfg.eot])


def test_repetition_penalty_and_ngram(cfg):
    o = _Scripted(cfg, None)
    lg = np.zeros(cfg.n_vocab, np.float32)
    lg[7], lg[8] = 2.0, -2.0
    lp = _proc(o, lg.copy(), [7, 8, 7], repetition_penalty=2.0)
    base = _proc(o, lg.copy(), [])
    assert lp[7] < base[7] and lp[8] < base[8]                 # 2.0 -> 1.0, -2.0 -> -4.0
    # no-repeat 3-gram: history ... a b c a b  -> c is banned
    lp = _proc(o, lg.copy(), [20, 21, 22, 20, 21], no_repeat_ngram_size=3)
    assert np.isneginf(lp[22]) and np.isfinite(lp[23])


def test_timestamp_rules(cfg):
    o = _Scripted(cfg, None)
    tb, V = cfg.timestamp_begin, cfg.n_vocab
    lg = np.zeros(V, np.float32)
    # first step: only timestamps <= tb + max_initial_timestamp_index
    lp = _proc(o, lg.copy(), [], with_timestamps=True, max_initial_timestamp_index=50)
    assert np.isneginf(lp[:tb]).all() and np.isfinite(lp[tb:tb + 51]).all() and np.isneginf(lp[tb + 51:]).all()
    # after <ts> text: timestamps may not decrease; no_timestamps always banned
    lg2 = lg.copy()
    lg2[:tb] += 10.0                                            # make text dominate rule (e): log(1501) < 10
    lp = _proc(o, lg2, [tb + 10, 30], with_timestamps=True)
    assert np.isneginf(lp[cfg.no_timestamps]) and np.isneginf(lp[tb:tb + 11]).all() and np.isfinite(lp[tb + 11])
    assert np.isfinite(lp[30])
    # after text <ts>: must be followed by a timestamp (pair) or eot -> text ids < eot banned
    lp = _proc(o, lg.copy(), [tb + 10, 30, tb + 20], with_timestamps=True)
    assert np.isneginf(lp[:cfg.eot]).all() and np.isfinite(lp[tb + 20]) and np.isneginf(lp[tb + 19])
    # after <ts><ts>: no third timestamp
    lp = _proc(o, lg2.copy(), [tb + 10, tb + 10], with_timestamps=True)
    assert np.isneginf(lp[tb:]).all() and np.isfinite(lp[30])
    # rule (e): if timestamp mass beats every single text token, text is masked
    lg3 = np.full(V, -20.0, np.float32)
    lg3[40] = 0.0                                               # best text logprob ~ log(1/(1+1501*e^-?))
    lg3[tb + 100:] = -1.0                                       # many timestamps, jointly heavier
    lp = _proc(o, lg3, [tb + 10, 30], with_timestamps=True)
    assert np.isneginf(lp[:tb]).all() and np.isclose(np.exp(lp[tb:][np.isfinite(lp[tb:])]).sum(), 1.0, atol=1e-4)


def test_budget_rule():
    assert max_new_tokens(448, 4) == 444 and max_new_tokens(104, 4) == 100 and max_new_tokens(3, 4) == 0


def test_greedy_and_beam_on_scripted_logits(cfg):
    V, eot = cfg.n_vocab, cfg.eot

    def table(tok, pos):
        lg = np.full(V, -10.0, np.float32)
        if pos < 6:
            lg[100 + pos] = 2.0           # greedy path: 103, 104, 105 (prompt has 4 tokens -> pos 3..)
            lg[200 + pos] = 1.9           # close runner-up
        else:
            lg[eot] = 5.0
        return lg
    o = _Scripted(cfg, table)
    prompt = [cfg.sot, cfg.lang_begin, cfg.transcribe, cfg.no_timestamps]
    enc = np.zeros((1, 1500, cfg.d_model), np.float32)
    g = o.generate(enc, [prompt], beam_size=1, max_length=20, suppress_blank=False)[0]
    assert g.sequences_ids[0] == [103, 104, 105]
    # score = cum / len^1 with cum including the eot log-prob; the reference recovers
    # avg_logprob = score * len / (len + 1)   (transcribe.py:241-246)
    lp_step = 2.0 - np.log(np.exp(2.0) + np.exp(1.9) + (V - 2) * np.exp(-10.0))
    lp_eot = 5.0 - np.log(np.exp(5.0) + (V - 1) * np.exp(-10.0))
    assert g.scores[0] == pytest.approx((3 * lp_step + lp_eot) / 3, abs=1e-4)
    b = o.generate(enc, [prompt], beam_size=3, max_length=20, suppress_blank=False, num_hypotheses=3)[0]
    assert b.sequences_ids[0] == [103, 104, 105]
    assert len(b.sequences_ids) == 3 and b.scores[0] >= b.scores[1] >= b.scores[2]
    assert b.sequences_ids[1] in ([203, 104, 105], [103, 204, 105], [103, 104, 205])
    # max-length finalisation: budget 2 -> live beams are returned without eot
    c = o.generate(enc, [prompt], beam_size=2, max_length=len(prompt) + 2, suppress_blank=False)[0]
    assert c.sequences_ids[0] == [103, 104]


def test_topk_median_dtw_helpers():
    x = np.array([1.0, 3.0, 3.0, 2.0, 3.0], np.float32)
    assert _topk_stable(x, 3).tolist() == [1, 2, 4]            # ties -> lowest index first
    m = _median_filter(np.array([[1.0, 9.0, 2.0, 8.0, 3.0]]), 3)
    assert m.tolist() == [[9.0, 2.0, 8.0, 3.0, 8.0]]            # reflect padding at both ends
    cost = 1.0 - 2.0 * np.eye(3)
    ti, fi = _dtw(cost)
    assert ti.tolist() == [0, 1, 2] and fi.tolist() == [0, 1, 2]
    # more frames than tokens: every token row is visited, the path is monotone and ends in the corner
    ti, fi = _dtw(np.array([[-1.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, -1.0]]))
    # (ties in the recurrence fall through to the "left" move, exactly like openai's dtw_cpu)
    assert ti.tolist() == [0, 0, 1, 1, 1] and fi.tolist() == [0, 1, 1, 2, 3]


def test_fully_forced_call_in_one_pass_equals_step_by_step():
    """OracleWhisper scores a GIVEN sequence (conftest.forced_result: every step forced) with ONE decoder_full pass over
    prompt + tokens instead of one decoder_step per token (a 224-step large-v3 score sweeps the weights once, not 224
    times).  Same layers, same rules, applied step by step to the precomputed logits.  In fp32 arithmetic the two paths
    agree to summation order (1e-5); with fp16 emulation a last-bit difference of a matmul can move an fp16 rounding, which
    is the evaluation-order noise the GPU tolerances already carry (bounded here at 2e-4 per token)."""
    from faster_whisper_amd import get_config, synthetic_weights
    from oracle.whisper import OracleWhisper
    from conftest import forced_result
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=3)
    enc = (np.random.default_rng(0).standard_normal((1500, cfg.d_model)) * 0.5).astype(np.float32)
    for mode, per_tok in ((dict(emulate_fp16=False), 1e-6), (dict(emulate_fp16=True), 2e-4)):
        o = OracleWhisper(cfg, w, **mode)
        for ts in (False, True):
            prompt = list(cfg.sot_sequence) + ([] if ts else [cfg.no_timestamps])
            kw = dict(max_length=len(prompt) + 20, suppress_blank=True, length_penalty=1.0, max_initial_timestamp_index=50)
            ids = o.generate(enc[None], [prompt], beam_size=1, **kw)[0].sequences_ids[0]
            for seq in (ids, ids[:7]):            # the whole budget; a shorter one (ends with <eot>)
                res = {}
                for bf in (False, True):
                    o.batch_forced = bf
                    res[bf] = forced_result(o, enc, prompt, seq, dict(kw, beam_size=1))
                a, b = res[False], res[True]
                n = len(seq) + (1 if len(seq) < 20 else 0)
                assert a.sequences_ids == b.sequences_ids
                assert abs(a.scores[0] - b.scores[0]) / n < per_tok, (mode, ts, a.scores, b.scores)
                assert abs(a.no_speech_prob - b.no_speech_prob) < 1e-7
                assert len(a.forced_gaps) == len(b.forced_gaps) == n and len(a.margins) == len(b.margins)
                if not mode["emulate_fp16"]:
                    assert max(abs(x - y) for x, y in zip(a.margins, b.margins)) < 1e-4
                    assert max(abs(x - y) for x, y in zip(a.forced_gaps, b.forced_gaps)) < 1e-4
