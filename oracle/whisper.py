"""CPU restatement of everything behind `ctranslate2.models.Whisper` (test oracle).

PARITY UNPINNED: CTranslate2 (`ctranslate2>=4.0,<5`, /root/reference/requirements.txt:1) is a
third-party dependency that is not vendored in /root/reference and not installed; this file
restates its published behaviour (OpenNMT/CTranslate2 4.x `src/models/whisper.cc`,
`src/layers/whisper.cc`, `src/decoding.cc`; openai/whisper `model.py`, `decoding.py`,
`timing.py`) and is anchored on the reference's call sites:
    encode            faster_whisper/transcribe.py:1391-1400
    generate          transcribe.py:222-236 (batched), :1433-1459 (sequential)
    score use         transcribe.py:241-246, :1463-1466  (score = cum_logprob / len^length_penalty,
                      len excludes <|endoftext|>; avg_logprob = cum / (len + 1))
    detect_language   transcribe.py:215, :1193, :1823-1828
    align             transcribe.py:1709-1746
The architecture is cross-checked against the installed `transformers` Whisper
(tests/test_oracle_arch.py), and so is — round 6 — the BEAM SEARCH bookkeeping: on the same weights
and rules, `transformers`' own beam search (`GenerationMixin.generate(num_beams=5, early_stopping=True,
length_penalty=0)`, an independent implementation of the top-2K / first-K-slots / secondary-candidate
scheme) returns the same five hypotheses in the same order with the same scores, with hypotheses
finishing before the budget in different steps (test_beam_search_matches_transformers_beam_search).
Every rule that is remembered rather than verified is tagged [CT2-ext].

Numerics: float32 torch on the CPU.  With `emulate_fp16=True` the tensors the GPU engine
stores in fp16 (weights, activations between kernels, attention probabilities) are rounded
to fp16 at the same points, so the remaining engine-vs-oracle difference is accumulation
order only.

`fold_ln` (attribute, fp16 only; default False) switches the DECODER to the engine's evaluation order of the same
model: a LayerNorm that feeds a linear is folded into it, y = rstd * (x W'^T - mu * rowsum(W')) + (W b + bias) with
W' = fp16(W * g), so the normalised rows are never rounded to fp16 but the gains are rounded into the weights.  Both
orders are valid fp16 evaluations of one model; at large-v3 size (32 layers) they differ by up to ~1.3e-3 per token
in log-prob and ~3e-3 in a language probability — the same size as fp16 vs fp32 (tests/numerics_ln_fold_noise.py),
which is why the full-size parity test compares the engine against THIS order tightly and against the explicit
order / fp32 only at that measured noise level.

`int8=True` restates compute_type "int8_float16" ([CT2-ext] CTranslate2 convention): every Dense
weight is quantised per output row (scale = 127 / absmax, round-half-even), every Dense input is
quantised per row the same way at run time, the product is accumulated exactly in integers and
de-quantised by (row scale x weight scale); convolutions, LayerNorm, attention and the residual
stream stay fp16.  The token embedding shares the quantised projection weight, so a lookup
returns the de-quantised row.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))


@dataclass
class GenResult:
    sequences_ids: List[List[int]]
    scores: List[float]
    no_speech_prob: float
    # diagnostics for margin-aware parity checks.  beam_size == 1: per generated step, (top1 - top2) of the processed
    # log-probs along the greedy path.  beam search: per step, the gap between candidates K and K + 1 of
    # cum + logp (the pruning boundary)
    margins: List[float] = field(default_factory=list)
    # teacher forcing (force_tokens, greedy): per step, log-prob of the best token minus log-prob of the FORCED token
    # (0 where the forced token is the arg-max): how far every choice of a given sequence is from greedy-optimal
    forced_gaps: List[float] = field(default_factory=list)
    # greedy / teacher forcing with timestamps: per step, log(timestamp mass) - max text log-prob — the quantity whose
    # SIGN decides timestamp rule (e) (text forbidden when the timestamps together outweigh the best text token).  A step
    # where it is within the numerical noise is a tied RULE: engine and oracle may renormalise differently there, and the
    # log-prob of the very same token differs by log(timestamp mass); nan where the rule does not apply
    rule_margins: List[float] = field(default_factory=list)


@dataclass
class AlignResult:
    alignments: List[tuple]
    text_token_probs: List[float]


def max_new_tokens(max_length: int, prompt_len: int) -> int:
    """Generated-token budget.  The reference treats `max_length` as prompt + new tokens
    (transcribe.py:193-207: max_length = len(prompt) + max_new_tokens), so the budget is
    max_length - len(prompt).  [CT2-ext] CTranslate2 may additionally cap at max_length // 2
    (openai `sample_len`); unverifiable offline — keep this the single place that decides."""
    return max(0, max_length - prompt_len)


class _Normed:
    """rows normalised by a decoder LayerNorm whose gain / bias are folded into the consuming linear (fold_ln)"""
    def __init__(self, z: torch.Tensor, p: str):
        self.z, self.p = z, p

    def squeeze(self, dim):
        return _Normed(self.z.squeeze(dim), self.p)

    def __getitem__(self, idx):
        return _Normed(self.z[idx], self.p)


class OracleWhisper:
    def __init__(self, cfg, weights: Dict[str, np.ndarray], emulate_fp16: bool = False, threads: Optional[int] = None,
                 int8: bool = False):
        if threads:
            torch.set_num_threads(threads)
        self.cfg = cfg
        self.h = emulate_fp16 or int8
        self.int8 = int8
        self.fold_ln = False      # see the module docstring; toggled by the full-size parity test
        self.batch_forced = True  # fully teacher-forced calls: one decoder_full pass instead of a step per token
        self._folded = {}
        self.w = {k: self._r(_t(v)) for k, v in weights.items()}
        d = cfg.d_model
        self.d, self.H = d, cfg.n_heads
        self.q = {}
        if int8:
            for k, v in self.w.items():
                if k.endswith(".w") and v.dim() == 2:
                    self.q[k] = self._quant_rows(v)
            wq, ws = self._quant_rows(self.w["dec.tok_emb"])
            self.q["dec.tok_emb"] = (wq, ws)
            self.w["dec.tok_emb"] = self._r(wq.float() * ws[:, None])

    @staticmethod
    def _quant_rows(x: torch.Tensor):
        """per-row symmetric int8: q = rint(x * (127 / absmax)), de-quantisation factor absmax / 127
        (float32 arithmetic, round-half-even; an all-zero row gets q = 0, factor 1)"""
        amax = x.abs().amax(dim=-1)
        one = torch.tensor(127.0, dtype=torch.float32)
        inv = torch.where(amax > 0, one / amax, torch.zeros_like(amax))
        ds = torch.where(amax > 0, amax / one, torch.ones_like(amax))
        q = torch.round(x * inv.unsqueeze(-1))            # the int8 codes, held as float32 (exact: |q| <= 127)
        return q, ds

    QK_BLOCK = 1024      # 127 * 127 * 1024 < 2^24: a float32 product over that many codes is an exact integer

    def _qmatmul(self, x: torch.Tensor, key: str) -> torch.Tensor:
        """int8_float16 Dense core: quantise the rows of x, exact integer product, de-quantise.
        The integer product runs on float32 BLAS over K blocks of 1024 (every partial sum of a block is an integer
        below 2^24, so each block is exact whatever the summation order); the blocks are added in float64 (exact:
        |acc| < 2^53) — the same integers an int64 product gives (tests/test_oracle_int8.py), at the speed of sgemm and
        without converting the weights on every call."""
        wq, ws = self.q[key]
        xq, xs = self._quant_rows(x)
        K = xq.shape[-1]
        acc = None
        for k0 in range(0, K, self.QK_BLOCK):
            part = torch.matmul(xq[..., k0:k0 + self.QK_BLOCK], wq[:, k0:k0 + self.QK_BLOCK].t()).double()
            acc = part if acc is None else acc + part
        return acc.float() * xs.unsqueeze(-1) * ws

    def _dense(self, x: torch.Tensor, key: str, bias_key: Optional[str]) -> torch.Tensor:
        if self.int8:
            y = self._qmatmul(x, key)
            return y + self.w[bias_key] if bias_key else y
        return torch.nn.functional.linear(x, self.w[key], self.w[bias_key] if bias_key else None)

    # fp16 rounding point
    def _r(self, x: torch.Tensor) -> torch.Tensor:
        return x.half().float() if self.h else x

    def _ln(self, x, p):
        if self.fold_ln and not self.int8 and p.startswith("dec."):
            mu = x.mean(-1, keepdim=True)
            var = ((x - mu) ** 2).mean(-1, keepdim=True)
            return _Normed((x - mu) * torch.rsqrt(var + 1e-5), p)
        return self._r(torch.nn.functional.layer_norm(x, (self.d,), self.w[p + ".g"], self.w[p + ".b"], 1e-5))

    def _fold(self, wkey: str, bkey: Optional[str], lnp: str):
        """(W' = fp16(W * g), cf = W b + bias) of linear `wkey` behind LayerNorm `lnp` (engine.hip packer: add_folded)"""
        k = (wkey, lnp)
        if k not in self._folded:
            W, g, b = self.w[wkey], self.w[lnp + ".g"], self.w[lnp + ".b"]
            cf = (W.double() @ b.double())
            if bkey:
                cf = cf + self.w[bkey].double()
            self._folded[k] = (self._r(W * g), cf.float())
        return self._folded[k]

    def _lin(self, x, p, act=False, res=None):
        if isinstance(x, _Normed):
            Wf, cf = self._fold(p + ".w", p + ".b", x.p)
            y = torch.matmul(x.z, Wf.t()) + cf
        else:
            y = self._dense(x, p + ".w", p + ".b")
        if act:
            y = torch.nn.functional.gelu(y)  # exact erf GELU
        if res is not None:
            y = y + res
        return self._r(y)

    def _heads(self, x):  # [B, T, d] -> [B, H, T, 64]
        B, T, _ = x.shape
        return x.view(B, T, self.H, 64).transpose(1, 2)

    def _attn(self, q, k, v, mask=None, return_probs=False):
        """q [B,H,Tq,64] (unscaled), k/v [B,H,Tk,64]; softmax(q k^T / 8) v"""
        if q.shape[0] > 1 and k.shape[0] == q.shape[0] and k.stride(0) == 0 and v.stride(0) == 0 and mask is None:
            # the rows share ONE K/V (the beams of a chunk: cross K/V expanded over the beam axis).  Same arithmetic
            # with the rows folded into the query axis — a batched matmul over the expanded view would first copy
            # K and V once per row (38 MB per layer and step at large-v3).
            R, H, Tq, _ = q.shape
            qf = q.permute(1, 0, 2, 3).reshape(1, H, R * Tq, 64)
            out = self._attn(qf, k[:1], v[:1], None, return_probs)
            o, p = out if return_probs else (out, None)
            o = o.reshape(R, Tq, H * 64)
            if return_probs:
                return o, p.reshape(H, R, Tq, -1).permute(1, 0, 2, 3)
            return o
        s = torch.matmul(q * 0.125, k.transpose(-1, -2))
        if mask is not None:
            s = s + mask
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(self._r(p), v)
        B, H, Tq, _ = o.shape
        o = self._r(o.transpose(1, 2).reshape(B, Tq, H * 64))
        return (o, p) if return_probs else o

    # ------------------------------------------------------------------ encoder
    def encode(self, features: np.ndarray, n_layers: Optional[int] = None) -> np.ndarray:
        """features float32 [B, n_mels, 3000] -> [B, 1500, d]
        n_layers: run only the first n transformer blocks (bench.py's bounded CPU-baseline sample)"""
        with torch.no_grad():
            x = self._r(_t(features))
            x = self._r(torch.nn.functional.gelu(
                torch.nn.functional.conv1d(x, self.w["enc.conv1.w"], self.w["enc.conv1.b"], padding=1)))
            x = torch.nn.functional.gelu(
                torch.nn.functional.conv1d(x, self.w["enc.conv2.w"], self.w["enc.conv2.b"], stride=2, padding=1))
            x = self._r(x.transpose(1, 2) + self.w["enc.pos"])
            for i in range(self.cfg.n_enc_layers if n_layers is None else n_layers):
                p = f"enc.{i}."
                xn = self._ln(x, p + "ln1")
                qkv = self._lin(xn, p + "attn.qkv")
                q, k, v = qkv.split(self.d, dim=-1)
                a = self._attn(self._heads(q), self._heads(k), self._heads(v))
                x = self._lin(a, p + "attn.out", res=x)
                xn = self._ln(x, p + "ln2")
                hdn = self._lin(xn, p + "ffn1", act=True)
                x = self._lin(hdn, p + "ffn2", res=x)
            x = self._ln(x, "enc.ln_post")
            return x.numpy()

    # ------------------------------------------------------------------ decoder
    def cross_kv(self, enc: torch.Tensor):
        out = []
        for i in range(self.cfg.n_dec_layers):
            p = f"dec.{i}.cross.kv"
            kv = self._r(self._dense(enc, p + ".w", p + ".b"))
            k, v = kv.split(self.d, dim=-1)
            out.append((self._heads(k), self._heads(v)))
        return out

    def decoder_full(self, tokens: torch.Tensor, ckv, return_cross_probs=False):
        """teacher-forced pass.  tokens [R, n] (long); ckv per layer ([R,H,1500,64], ...).
        Returns hidden [R, n, d] after the final LN (and per-layer cross-attn probs)."""
        R, n = tokens.shape
        x = self._r(self.w["dec.tok_emb"][tokens] + self.w["dec.pos"][:n])
        causal = torch.full((n, n), float("-inf")).triu(1)
        probs = []
        for i in range(self.cfg.n_dec_layers):
            p = f"dec.{i}."
            xn = self._ln(x, p + "ln1")
            q, k, v = self._lin(xn, p + "self.qkv").split(self.d, dim=-1)
            a = self._attn(self._heads(q), self._heads(k), self._heads(v), mask=causal)
            x = self._lin(a, p + "self.out", res=x)
            xn = self._ln(x, p + "ln2")
            q = self._lin(xn, p + "cross.q")
            ck, cv = ckv[i]
            if return_cross_probs:
                a, pr = self._attn(self._heads(q), ck, cv, return_probs=True)
                probs.append(pr)
            else:
                a = self._attn(self._heads(q), ck, cv)
            x = self._lin(a, p + "cross.out", res=x)
            xn = self._ln(x, p + "ln3")
            x = self._lin(self._lin(xn, p + "ffn1", act=True), p + "ffn2", res=x)
        x = self._ln(x, "dec.ln")
        return (x, probs) if return_cross_probs else x

    def logits(self, hidden) -> torch.Tensor:
        if isinstance(hidden, _Normed):
            Wf, cf = self._fold("dec.tok_emb", None, hidden.p)
            return torch.matmul(hidden.z, Wf.t()) + cf
        return self._dense(hidden, "dec.tok_emb", None)

    class _Cache:
        """self-attention K/V per layer for R rows, positions filled so far"""
        def __init__(self, n_layers):
            self.k = [None] * n_layers
            self.v = [None] * n_layers

        def reorder(self, idx: torch.Tensor):
            self.k = [t.index_select(0, idx) for t in self.k]
            self.v = [t.index_select(0, idx) for t in self.v]

    def decoder_step(self, tok: torch.Tensor, pos: int, cache: "_Cache", ckv):
        """one position for R rows with KV cache.  tok [R] long -> hidden [R, d]"""
        x = self._r(self.w["dec.tok_emb"][tok] + self.w["dec.pos"][pos]).unsqueeze(1)
        for i in range(self.cfg.n_dec_layers):
            p = f"dec.{i}."
            xn = self._ln(x, p + "ln1")
            q, k, v = self._lin(xn, p + "self.qkv").split(self.d, dim=-1)
            k, v = self._heads(k), self._heads(v)
            cache.k[i] = k if cache.k[i] is None else torch.cat([cache.k[i], k], dim=2)
            cache.v[i] = v if cache.v[i] is None else torch.cat([cache.v[i], v], dim=2)
            a = self._attn(self._heads(q), cache.k[i], cache.v[i])
            x = self._lin(a, p + "self.out", res=x)
            xn = self._ln(x, p + "ln2")
            q = self._lin(xn, p + "cross.q")
            ck, cv = ckv[i]
            a = self._attn(self._heads(q), ck, cv)
            x = self._lin(a, p + "cross.out", res=x)
            xn = self._ln(x, p + "ln3")
            x = self._lin(self._lin(xn, p + "ffn1", act=True), p + "ffn2", res=x)
        return self._ln(x, "dec.ln").squeeze(1)

    # ------------------------------------------------------------------ logits rules
    def _process_logits(self, logits: np.ndarray, generated: List[int], with_timestamps: bool, suppress_mask,
                        suppress_blank: bool, max_initial_timestamp_index: int, repetition_penalty: float,
                        no_repeat_ngram_size: int, min_new_tokens: int) -> np.ndarray:
        """one row: raw logits [V] float32 -> processed log-probs [V] float32."""
        c = self.cfg
        lg = logits.astype(np.float32).copy()
        NEG = np.float32(-np.inf)
        n = len(generated)
        if repetition_penalty != 1.0 and n:                       # CT2 RepetitionPenalty
            ids = np.unique(np.asarray(generated))
            v = lg[ids]
            lg[ids] = np.where(v < 0, v * np.float32(repetition_penalty), v / np.float32(repetition_penalty))
        if no_repeat_ngram_size > 0 and n + 1 >= no_repeat_ngram_size:   # CT2 NoRepeatNgram
            k = no_repeat_ngram_size
            prefix = tuple(generated[n - (k - 1):]) if k > 1 else ()
            for s in range(0, n - k + 1):
                if tuple(generated[s:s + k - 1]) == prefix:
                    lg[generated[s + k - 1]] = NEG
        if suppress_blank and n == 0:                             # SuppressTokensBegin
            for t in c.suppress_begin:
                lg[t] = NEG
        if suppress_mask is not None:                             # SuppressTokens
            lg[suppress_mask] = NEG
        if n < min_new_tokens:                                    # benchmark-only control
            lg[c.eot] = NEG
        if with_timestamps:                                       # ApplyTimestampRules
            tb = c.timestamp_begin
            lg[c.no_timestamps] = NEG
            last_ts = n >= 1 and generated[-1] >= tb
            penult_ts = n < 2 or generated[-2] >= tb
            if last_ts:
                if penult_ts:
                    lg[tb:] = NEG
                else:
                    lg[:c.eot] = NEG
            ts = [t for t in generated if t >= tb]
            if ts:
                last = ts[-1] if (last_ts and not penult_ts) else ts[-1] + 1
                lg[tb:last] = NEG
            if n == 0:
                lg[:tb] = NEG
                if max_initial_timestamp_index is not None and max_initial_timestamp_index >= 0:
                    lg[tb + max_initial_timestamp_index + 1:] = NEG
            lp = _log_softmax(lg)
            ts_lp = _logsumexp(lp[tb:])
            text_max = lp[:tb].max()
            self._rule_margin = float(ts_lp - text_max) if np.isfinite(ts_lp) and np.isfinite(text_max) else float("nan")
            if ts_lp > text_max:
                lg[:tb] = NEG
        else:
            self._rule_margin = float("nan")
        return _log_softmax(lg)

    # ------------------------------------------------------------------ generate
    def generate(self, enc: np.ndarray, prompts: Sequence[Sequence[int]], beam_size=5, patience=1.0,
                 num_hypotheses=1, length_penalty=1.0, repetition_penalty=1.0, no_repeat_ngram_size=0,
                 max_length=448, return_scores=True, return_no_speech_prob=True, max_initial_timestamp_index=50,
                 suppress_blank=True, suppress_tokens=None, sampling_topk=1, sampling_temperature=1.0,
                 min_new_tokens=0, force_tokens: Optional[Sequence[Sequence[int]]] = None,
                 seed: int = 0) -> List[GenResult]:
        """CTranslate2 Whisper.generate semantics (greedy for beam_size == 1, beam search otherwise).
        force_tokens: teacher forcing for margin diagnostics (greedy only): the chosen token at each
        step is taken from this list instead of the argmax."""
        c = self.cfg
        enc_t = self._r(_t(enc))
        sup = None
        if suppress_tokens is not None:
            ids = [t for t in suppress_tokens if 0 <= t < c.n_vocab]
            if ids:
                sup = np.asarray(sorted(set(ids)), dtype=np.int64)
        out = []
        sampling = beam_size == 1 and sampling_topk != 1
        with torch.no_grad():
            for b, prompt in enumerate(prompts):
                if sampling:
                    # random sampling = num_hypotheses independent beam-1 chunks (rows b*nh + j), best first
                    hyps = []
                    for j in range(num_hypotheses):
                        smp = (seed, b * num_hypotheses + j, 1.0 / sampling_temperature)
                        r = self._generate_one(enc_t[b:b + 1], list(prompt), 1, patience, 1, length_penalty,
                                               repetition_penalty, no_repeat_ngram_size, max_length,
                                               max_initial_timestamp_index, suppress_blank, sup, min_new_tokens,
                                               None, sample=smp)
                        hyps.append(r)
                    order = sorted(range(len(hyps)), key=lambda t: -hyps[t].scores[0])
                    out.append(GenResult([hyps[t].sequences_ids[0] for t in order], [hyps[t].scores[0] for t in order],
                                         hyps[0].no_speech_prob))
                    continue
                out.append(self._generate_one(enc_t[b:b + 1], list(prompt), beam_size, patience, num_hypotheses,
                                              length_penalty, repetition_penalty, no_repeat_ngram_size, max_length,
                                              max_initial_timestamp_index, suppress_blank, sup, min_new_tokens,
                                              force_tokens[b] if force_tokens is not None else None))
        return out

    def _generate_one(self, enc1, prompt, K, patience, num_hyp, lp_pow, rep_pen, ngram, max_length, mits,
                      suppress_blank, sup, min_new, forced, sample=None):
        c = self.cfg
        P = len(prompt)
        budget = max_new_tokens(max_length, P)
        with_ts = c.no_timestamps not in prompt
        ckv1 = self.cross_kv(enc1)
        sot_pos = max((i for i, t in enumerate(prompt) if t == c.sot), default=-1)
        # ---- a fully teacher-forced greedy call (every step's token is given: tests score a hypothesis) needs no
        #      autoregression: ONE pass of decoder_full over prompt + forced tokens yields the logits of every step — the
        #      same layers, the same rounding points, the same rules applied step by step below; what differs from the
        #      position-by-position path is fp32 summation order inside BLAS (tests/test_oracle_decode.py pins the two
        #      against each other).  A 224-step large-v3 score: one sweep over the weights instead of 224.
        full = (forced is not None and K == 1 and sample is None and self.batch_forced and budget > 0 and
                (len(forced) >= budget or (len(forced) > 0 and forced[-1] == c.eot)))
        if full:
            m = min(len(forced), budget)
            toks = torch.tensor([list(prompt) + [int(t) for t in forced[:m - 1]]], dtype=torch.long)
            hid = self.decoder_full(toks, ckv1)
            no_speech = 0.0
            if sot_pos >= 0:
                no_speech = float(torch.softmax(self.logits(hid[:, sot_pos])[0], dim=-1)[c.no_speech])
            all_logits = self.logits(hid[0, P - 1:P - 1 + m]).numpy()

            def proc(lg_row, gen):
                return self._process_logits(lg_row, gen, with_ts, sup, suppress_blank, mits, rep_pen, ngram, min_new)

            return self._greedy(None, ckv1, all_logits[0:1], proc, P, budget, lp_pow, no_speech, forced, None,
                                step_logits=all_logits)
        # ---- prompt forward (all but the last token), no_speech at the <sot> position
        cache = self._Cache(c.n_dec_layers)
        no_speech = 0.0
        for pos in range(P - 1):
            h = self.decoder_step(torch.tensor([prompt[pos]]), pos, cache, ckv1)
            if pos == sot_pos:
                pr = torch.softmax(self.logits(h)[0], dim=-1)
                no_speech = float(pr[c.no_speech])
        # ---- first real step: the last prompt token, one row
        h = self.decoder_step(torch.tensor([prompt[P - 1]]), P - 1, cache, ckv1)
        if sot_pos == P - 1:
            no_speech = float(torch.softmax(self.logits(h)[0], dim=-1)[c.no_speech])
        logits = self.logits(h).numpy()
        if budget == 0:
            return GenResult([[]], [0.0], no_speech)

        def proc(lg_row, gen):
            return self._process_logits(lg_row, gen, with_ts, sup, suppress_blank, mits, rep_pen, ngram, min_new)

        if K == 1:
            return self._greedy(cache, ckv1, logits, proc, P, budget, lp_pow, no_speech, forced, sample)

        # ---------------- beam search [CT2-ext: BeamSearch::search in src/decoding.cc] ----------------
        ckv = [(k.expand(K, -1, -1, -1), v.expand(K, -1, -1, -1)) for k, v in ckv1]
        max_fin = max(1, int(round(K * patience)))
        beams = [[] for _ in range(K)]          # generated tokens per live beam
        cum = np.zeros(K, dtype=np.float32)
        finished = []                           # (score, tokens, cum)
        beam_gaps = []
        n_live_src = 1                          # first step expands from beam 0 only
        V = c.n_vocab
        step = 0
        cache.reorder(torch.zeros(K, dtype=torch.long))
        while True:
            # candidates: top 2K of cum[k] + logp[k][v], ties -> lowest flat index
            lp = np.stack([proc(logits[k], beams[k]) for k in range(n_live_src)])
            flat = (cum[:n_live_src, None] + lp).reshape(-1)
            order = _topk_stable(flat, 2 * K)
            # pruning-boundary gap of this step (candidate K vs K + 1): an engine whose scores carry noise of that
            # size may legitimately keep a different beam set from here on (tests: conftest.check_hypothesis)
            if len(order) > K and np.isfinite(flat[order[K]]):
                beam_gaps.append(float(flat[order[K - 1]] - flat[order[K]]))
            last_step = (step + 1) >= budget
            new_beams, new_cum, parents, new_tok = [], [], [], []
            sec = K                              # secondary candidate cursor
            for slot in range(K):
                j = slot
                cand = order[j] if j < len(order) else None
                if cand is not None and np.isfinite(flat[cand]) and (cand % V == c.eot or last_step):
                    kk, vv = divmod(int(cand), V)
                    toks = beams[kk] + ([] if vv == c.eot else [vv])
                    finished.append(_hyp(flat[cand], toks, lp_pow))
                    if last_step:
                        continue
                    # replace by the next non-eot secondary candidate
                    while sec < len(order) and order[sec] % V == c.eot:
                        sec += 1
                    j = sec
                    sec += 1
                    cand = order[j] if j < len(order) else None
                if cand is None:
                    continue
                kk, vv = divmod(int(cand), V)
                new_beams.append(beams[kk] + [vv])
                new_cum.append(flat[cand])
                parents.append(kk)
                new_tok.append(vv)
            step += 1
            if last_step or len(finished) >= max_fin or not new_beams:
                break
            while len(new_beams) < K:           # degenerate (fewer than K finite candidates): pad with dead beams
                new_beams.append(list(new_beams[0])); new_cum.append(np.float32(-np.inf))
                parents.append(parents[0]); new_tok.append(new_tok[0])
            beams, cum = new_beams, np.asarray(new_cum, dtype=np.float32)
            cache.reorder(torch.tensor(parents, dtype=torch.long))
            h = self.decoder_step(torch.tensor(new_tok), P - 1 + step, cache, ckv)
            logits = self.logits(h).numpy()
            n_live_src = K
        finished.sort(key=lambda t: -t[0])      # stable: earlier-finished first among equal scores
        best = finished[:max(1, num_hyp)]
        return GenResult([t[1] for t in best], [float(t[0]) for t in best], no_speech, margins=beam_gaps)

    def _greedy(self, cache, ckv1, logits, proc, P, budget, lp_pow, no_speech, forced, sample=None, step_logits=None):
        c = self.cfg
        gen, margins, gaps, rule_margins = [], [], [], []
        cum = np.float32(0.0)
        step = 0
        ended = False
        while step < budget:
            lp = proc(logits[0], gen)
            rule_margins.append(getattr(self, "_rule_margin", float("nan")))
            if sample is not None:   # Gumbel-max draw from softmax(lp / T); the score keeps the plain lp
                seed, row, inv_t = sample
                key = np.where(np.isfinite(lp), lp * np.float32(inv_t) + _gumbel(seed, row, step, lp.shape[0]),
                               np.float32(-np.inf)).astype(np.float32)
                order = _topk_stable(key, 2)
                margins.append(float(key[order[0]] - key[order[1]]))
            else:
                order = _topk_stable(lp, 2)
                margins.append(float(lp[order[0]] - lp[order[1]]))
            tok = int(order[0])
            if forced is not None and step < len(forced):
                tok = int(forced[step])
                gaps.append(float(lp[order[0]] - lp[tok]))
            cum = np.float32(cum + lp[tok])
            step += 1
            if tok == c.eot:
                ended = True
                break
            gen.append(tok)
            if step >= budget:
                break
            if step_logits is not None:     # fully forced: the logits of every step were computed in one pass
                if step >= step_logits.shape[0]:
                    break
                logits = step_logits[step:step + 1]
                continue
            h = self.decoder_step(torch.tensor([tok]), P - 1 + step, cache, ckv1)
            logits = self.logits(h).numpy()
        score = _hyp(cum, gen, lp_pow)[0]
        return GenResult([gen], [float(score)], no_speech, margins, gaps, rule_margins)

    # ------------------------------------------------------------------ detect_language
    def detect_language(self, enc: np.ndarray):
        """one decoder step on [sot]; softmax over the language ids only; sorted descending
        (ties: lower id first).  Returns per row a list of (token_id, prob)."""
        c = self.cfg
        with torch.no_grad():
            enc_t = self._r(_t(enc))
            ckv = self.cross_kv(enc_t)
            B = enc_t.shape[0]
            cache = self._Cache(c.n_dec_layers)
            h = self.decoder_step(torch.full((B,), c.sot, dtype=torch.long), 0, cache, ckv)
            lg = self.logits(h)[:, c.lang_begin:c.lang_begin + c.n_langs]
            pr = torch.softmax(lg, dim=-1).numpy()
        out = []
        for b in range(B):
            order = _topk_stable(pr[b], c.n_langs)
            out.append([(int(c.lang_begin + i), float(pr[b, i])) for i in order])
        return out

    # ------------------------------------------------------------------ align
    def alignment_heads(self):
        c = self.cfg
        if c.alignment_heads:
            return [tuple(x) for x in c.alignment_heads]
        # [CT2-ext] default: every head of the upper half of the decoder
        return [(l, h) for l in range(c.n_dec_layers // 2, c.n_dec_layers) for h in range(c.n_heads)]

    def align(self, enc: np.ndarray, start_sequence: Sequence[int], text_tokens: Sequence[Sequence[int]],
              num_frames, median_filter_width: int = 7) -> List[AlignResult]:
        """openai-whisper timing.find_alignment / CTranslate2 Whisper.align."""
        c = self.cfg
        B = enc.shape[0]
        nfs = [num_frames] * B if isinstance(num_frames, int) else list(num_frames)
        heads = self.alignment_heads()
        out = []
        with torch.no_grad():
            enc_t = self._r(_t(enc))
            for b in range(B):
                text = list(text_tokens[b])
                toks = list(start_sequence) + [c.no_timestamps] + text + [c.eot]
                ckv = self.cross_kv(enc_t[b:b + 1])
                hidden, probs = self.decoder_full(torch.tensor([toks]), ckv, return_cross_probs=True)
                n0 = len(start_sequence) + 1
                lg = self.logits(hidden[0, n0 - 1:n0 - 1 + len(text)])
                tp = torch.softmax(lg, dim=-1)
                text_probs = [float(tp[i, t]) for i, t in enumerate(text)]
                nfr = min(c.n_audio_ctx, max(1, nfs[b] // 2))
                w = torch.stack([probs[l][0, h] for (l, h) in heads])[:, :, :nfr]   # [heads, tokens, frames]
                mean = w.mean(dim=-2, keepdim=True)
                std = w.std(dim=-2, keepdim=True, unbiased=False)
                w = (w - mean) / std
                w = _median_filter(w.numpy(), median_filter_width)
                # rows <|notimestamps|> .. last text token (n_text + 1 rows; the attention at a position times
                # the token it predicts), openai-whisper timing.py `matrix[len(sot_sequence):-1]`; the reference
                # indexes the token jumps up to n_text (transcribe.py:1741-1745), which needs exactly these rows
                m = w.mean(axis=0)[n0 - 1:-1]
                ti, fi = _dtw(-m.astype(np.float64))
                out.append(AlignResult(list(zip(ti.tolist(), fi.tolist())), text_probs))
        return out


# ---------------------------------------------------------------------- helpers
def _log_softmax(x: np.ndarray) -> np.ndarray:
    m = x.max()
    if not np.isfinite(m):
        return np.full_like(x, -np.inf)
    e = np.exp((x - m).astype(np.float32), dtype=np.float32)
    return (x - m - np.log(e.sum(dtype=np.float32))).astype(np.float32)


def _logsumexp(x: np.ndarray) -> np.float32:
    m = x.max()
    if not np.isfinite(m):
        return np.float32(-np.inf)
    return np.float32(m + np.log(np.exp((x - m).astype(np.float32)).sum(dtype=np.float32)))


def _gumbel(seed: int, row: int, step: int, n: int) -> np.ndarray:
    """counter-based Gumbel noise, same integer hash as dec_kernels.hip::gumbel_noise"""
    M = np.uint64(0xFFFFFFFF)
    v = np.arange(n, dtype=np.uint64)
    h = (np.uint64(seed & 0xFFFFFFFF) ^ ((np.uint64(row) * np.uint64(0x9E3779B9)) & M)
         ^ ((np.uint64(step) * np.uint64(0x85EBCA6B)) & M) ^ ((v * np.uint64(0xC2B2AE35)) & M)
         ^ ((np.uint64((seed >> 32) & 0xFFFFFFFF) * np.uint64(0x27D4EB2F)) & M))
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    # 23 bits keep u strictly inside (0, 1) in float32 (see gumbel_noise in dec_kernels.hip)
    u = ((h >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def _topk_stable(x: np.ndarray, k: int) -> np.ndarray:
    """indices of the k largest values, descending; ties -> lowest index first"""
    k = min(k, x.shape[0])
    idx = np.argsort(-x, kind="stable")[:k]
    return idx


def _hyp(cum, toks, lp_pow):
    n = max(1, len(toks))
    score = np.float32(cum) / np.float32(n ** lp_pow) if lp_pow != 0 else np.float32(cum)
    return (float(score), list(toks), float(cum))


def _median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """median filter along the last axis with reflect padding (openai timing.median_filter)"""
    pad = width // 2
    if pad == 0 or x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def _dtw(cost: np.ndarray):
    """openai timing.dtw_cpu: D[i,j] = cost + min(diag, up, left); ties prefer diag, then up (i-1), then left."""
    N, M = cost.shape
    D = np.full((N + 1, M + 1), np.inf)
    tr = -np.ones((N + 1, M + 1), dtype=np.int8)
    D[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = D[i - 1, j - 1], D[i - 1, j], D[i, j - 1]
            if c0 < c1 and c0 < c2:
                cc, t = c0, 0
            elif c1 < c0 and c1 < c2:
                cc, t = c1, 1
            else:
                cc, t = c2, 2
            D[i, j] = cost[i - 1, j - 1] + cc
            tr[i, j] = t
    i, j = N, M
    tr[0, :] = 2
    tr[:, 0] = 1
    path = []
    while i > 0 or j > 0:
        path.append((i - 1, j - 1))
        t = tr[i, j]
        if t == 0:
            i -= 1; j -= 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    path = np.array(path[::-1])
    return path[:, 0], path[:, 1]
