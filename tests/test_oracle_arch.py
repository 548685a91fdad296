"""Architecture cross-check of the CPU oracle against an independent implementation of the
same network: the installed `transformers` WhisperForConditionalGeneration loaded with the
SAME synthetic weights (the only executable second opinion in this
environment).  Confirms layer order, pre-norm placement, exact-erf GELU, bias-free key
projection, [sin|cos] encoder positions, learned decoder positions, tied output embedding."""
import numpy as np
import pytest
import torch

from faster_whisper_amd import get_config, synthetic_weights
from oracle.whisper import OracleWhisper

transformers = pytest.importorskip("transformers")


def _load_into_hf(cfg, w):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    hc = WhisperConfig(vocab_size=cfg.n_vocab, num_mel_bins=cfg.n_mels, encoder_layers=cfg.n_enc_layers,
                       encoder_attention_heads=cfg.n_heads, decoder_layers=cfg.n_dec_layers,
                       decoder_attention_heads=cfg.n_heads, decoder_ffn_dim=4 * cfg.d_model,
                       encoder_ffn_dim=4 * cfg.d_model, d_model=cfg.d_model, max_source_positions=cfg.n_audio_ctx,
                       max_target_positions=cfg.n_text_ctx, pad_token_id=cfg.eot, bos_token_id=cfg.eot,
                       eos_token_id=cfg.eot, decoder_start_token_id=cfg.sot, activation_function="gelu",
                       dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    model = WhisperForConditionalGeneration(hc).eval()
    sd = {}
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    d = cfg.d_model

    def attn(prefix, qkv_w, qkv_b, out_w, out_b):
        sd[prefix + "q_proj.weight"], sd[prefix + "k_proj.weight"], sd[prefix + "v_proj.weight"] = t(qkv_w).split(d)
        qb, _, vb = t(qkv_b).split(d)
        sd[prefix + "q_proj.bias"], sd[prefix + "v_proj.bias"] = qb, vb
        sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"] = t(out_w), t(out_b)

    sd["model.encoder.conv1.weight"], sd["model.encoder.conv1.bias"] = t(w["enc.conv1.w"]), t(w["enc.conv1.b"])
    sd["model.encoder.conv2.weight"], sd["model.encoder.conv2.bias"] = t(w["enc.conv2.w"]), t(w["enc.conv2.b"])
    sd["model.encoder.embed_positions.weight"] = t(w["enc.pos"])
    for i in range(cfg.n_enc_layers):
        p, q = f"model.encoder.layers.{i}.", f"enc.{i}."
        attn(p + "self_attn.", w[q + "attn.qkv.w"], w[q + "attn.qkv.b"], w[q + "attn.out.w"], w[q + "attn.out.b"])
        sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = t(w[q + "ffn1.w"]), t(w[q + "ffn1.b"])
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = t(w[q + "ffn2.w"]), t(w[q + "ffn2.b"])
    sd["model.encoder.layer_norm.weight"], sd["model.encoder.layer_norm.bias"] = t(w["enc.ln_post.g"]), t(w["enc.ln_post.b"])
    sd["model.decoder.embed_tokens.weight"] = t(w["dec.tok_emb"])
    sd["model.decoder.embed_positions.weight"] = t(w["dec.pos"])
    for i in range(cfg.n_dec_layers):
        p, q = f"model.decoder.layers.{i}.", f"dec.{i}."
        attn(p + "self_attn.", w[q + "self.qkv.w"], w[q + "self.qkv.b"], w[q + "self.out.w"], w[q + "self.out.b"])
        sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        kvw, kvb = t(w[q + "cross.kv.w"]), t(w[q + "cross.kv.b"])
        sd[p + "encoder_attn.q_proj.weight"], sd[p + "encoder_attn.q_proj.bias"] = t(w[q + "cross.q.w"]), t(w[q + "cross.q.b"])
        sd[p + "encoder_attn.k_proj.weight"], sd[p + "encoder_attn.v_proj.weight"] = kvw.split(d)
        sd[p + "encoder_attn.v_proj.bias"] = kvb.split(d)[1]
        sd[p + "encoder_attn.out_proj.weight"], sd[p + "encoder_attn.out_proj.bias"] = t(w[q + "cross.out.w"]), t(w[q + "cross.out.b"])
        sd[p + "encoder_attn_layer_norm.weight"], sd[p + "encoder_attn_layer_norm.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = t(w[q + "ln3.g"]), t(w[q + "ln3.b"])
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = t(w[q + "ffn1.w"]), t(w[q + "ffn1.b"])
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = t(w[q + "ffn2.w"]), t(w[q + "ffn2.b"])
    sd["model.decoder.layer_norm.weight"], sd["model.decoder.layer_norm.bias"] = t(w["dec.ln.g"]), t(w["dec.ln.b"])
    sd["proj_out.weight"] = t(w["dec.tok_emb"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith("k_proj.bias") for m in missing), missing   # Whisper has no key bias
    return model


def test_oracle_matches_transformers_whisper():
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=3, dtype=np.float32)
    hf = _load_into_hf(cfg, w)
    oracle = OracleWhisper(cfg, w, emulate_fp16=False)
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((2, cfg.n_mels, 3000)).astype(np.float32) * 0.5
    tokens = torch.tensor([[cfg.sot, cfg.lang_begin, cfg.transcribe, 11, 12, 13], [cfg.sot, cfg.lang_begin + 1,
                                                                                  cfg.transcribe, 50, 60, 70]])
    with torch.no_grad():
        enc_hf = hf.model.encoder(torch.from_numpy(feats)).last_hidden_state.numpy()
        logits_hf = hf(input_features=torch.from_numpy(feats), decoder_input_ids=tokens).logits.numpy()
    enc = oracle.encode(feats)
    assert np.abs(enc - enc_hf).max() < 2e-4 * max(1.0, np.abs(enc_hf).max())
    with torch.no_grad():
        ckv = oracle.cross_kv(torch.from_numpy(enc))
        hidden = oracle.decoder_full(tokens, ckv)
        logits = oracle.logits(hidden).numpy()
    assert np.abs(logits - logits_hf).max() < 5e-4 * max(1.0, np.abs(logits_hf).max())
    # the KV-cached step path of the oracle equals its own teacher-forced pass
    with torch.no_grad():
        cache = oracle._Cache(cfg.n_dec_layers)
        for pos in range(tokens.shape[1]):
            h = oracle.decoder_step(tokens[:, pos], pos, cache, ckv)
        step_logits = oracle.logits(h).numpy()
    assert np.abs(step_logits - logits[:, -1]).max() < 1e-4


def test_greedy_decode_matches_a_loop_built_from_transformers_parts():
    """oracle.generate (greedy, timestamps on: prompt forward, KV-cached steps, suppress / blank / timestamp rules,
    log-softmax, score = cum / len) against a plain loop made of independent parts: transformers' Whisper forward +
    transformers' own logits processors.  Same tokens, same cumulative log-probability, same no-speech probability."""
    lp = pytest.importorskip("transformers.generation.logits_process")
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=9, dtype=np.float32)
    hf = _load_into_hf(cfg, w)
    oracle = OracleWhisper(cfg, w, emulate_fp16=False)
    rng = np.random.default_rng(4)
    feats = rng.standard_normal((2, cfg.n_mels, 3000)).astype(np.float32) * 0.5
    prompt = list(cfg.sot_sequence)                       # timestamps enabled
    sup = sorted({cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe, 3, 4, 5})
    n_new, mits = 14, 50

    class G:
        eos_token_id = cfg.eot
        bos_token_id = cfg.eot
        no_timestamps_token_id = cfg.no_timestamps
        max_initial_timestamp_index = mits
        _detect_timestamp_from_logprob = True
    procs = [lp.SuppressTokensAtBeginLogitsProcessor(list(cfg.suppress_begin), begin_index=len(prompt)),
             lp.SuppressTokensLogitsProcessor(sup),
             lp.WhisperTimeStampLogitsProcessor(G(), begin_index=len(prompt))]
    enc = oracle.encode(feats)
    got = oracle.generate(enc, [prompt] * 2, beam_size=1, max_length=len(prompt) + n_new, suppress_tokens=sup,
                          suppress_blank=True, max_initial_timestamp_index=mits, length_penalty=1.0)
    with torch.no_grad():
        for b in range(2):
            ids = torch.tensor([prompt])
            cum, toks, no_speech = 0.0, [], None
            for step in range(n_new):
                logits = hf(input_features=torch.from_numpy(feats[b:b + 1]), decoder_input_ids=ids).logits[0]
                if no_speech is None:   # softmax at the <|startoftranscript|> position
                    no_speech = float(torch.softmax(logits[0].float(), -1)[cfg.no_speech])
                scores = logits[-1:].float().clone()
                for p in procs:
                    scores = p(ids, scores)
                logp = torch.log_softmax(scores, dim=-1)[0]
                t = int(torch.argmax(logp))
                cum += float(logp[t])
                if t == cfg.eot:
                    break
                toks.append(t)
                ids = torch.cat([ids, torch.tensor([[t]])], dim=1)
            g = got[b]
            assert g.sequences_ids[0] == toks, (b, g.sequences_ids[0], toks)
            assert abs(g.scores[0] - cum / max(1, len(toks))) < 1e-4 * max(1.0, abs(cum))
            assert abs(g.no_speech_prob - no_speech) < 1e-5


@pytest.mark.parametrize("seed,eot_lift", [(21, 0.0), (23, 2.5), (27, 2.5), (28, 2.75), (24, 3.0), (26, 4.0)])
def test_beam_search_matches_transformers_beam_search(seed, eot_lift):
    """oracle beam search (the restatement of CTranslate2's BeamSearch::search: top 2K of cum + logp, the first K slots examined,
    an <eot> candidate among them becomes a finished hypothesis and its slot goes to the next non-<eot> candidate, stop at K
    finished hypotheses or at the length budget, best first) against transformers' OWN beam search on the same weights —
    `GenerationMixin.generate(num_beams=5, early_stopping=True)`, an independent implementation of that scheme — with the
    logits rules in CTranslate2's order (mask, THEN log-softmax: transformers masks after the log-softmax, so a re-normalising
    processor closes its list).  length_penalty = 0 on both sides: transformers counts the <eot> in a hypothesis' length,
    CTranslate2 does not (transcribe.py:241-246 inverts exactly that), so only the un-normalised score is comparable.
    `eot_lift` raises the <eot> logit through the final LayerNorm's bias so that hypotheses finish BEFORE the budget — in
    different steps, some below rank K (ignored), some replaced by secondary candidates."""
    from transformers import GenerationConfig
    from transformers.generation.logits_process import (LogitsProcessor, LogitsProcessorList, SuppressTokensAtBeginLogitsProcessor,
                                                        SuppressTokensLogitsProcessor)
    from transformers.generation.utils import GenerationMixin
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=seed, dtype=np.float32)
    if eot_lift:
        e = np.asarray(w["dec.tok_emb"][cfg.eot], dtype=np.float32)
        w["dec.ln.b"] = (np.asarray(w["dec.ln.b"], np.float32) + eot_lift * e / float(e @ e)).astype(np.float32)
    hf = _load_into_hf(cfg, w)
    oracle = OracleWhisper(cfg, w, emulate_fp16=False)
    rng = np.random.default_rng(seed)
    B, K, n_new = 3, 5, 14
    feats = rng.standard_normal((B, cfg.n_mels, 3000)).astype(np.float32) * 0.5
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = sorted({cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe, 3, 4, 5})

    class Renormalise(LogitsProcessor):          # CTranslate2: rules on the logits, then log-softmax
        def __call__(self, input_ids, scores):
            return torch.log_softmax(scores, dim=-1)

    procs = LogitsProcessorList([SuppressTokensAtBeginLogitsProcessor(list(cfg.suppress_begin), begin_index=len(prompt)),
                                 SuppressTokensLogitsProcessor(sup), Renormalise()])
    gc = GenerationConfig(num_beams=K, early_stopping=True, length_penalty=0.0, do_sample=False, num_return_sequences=K,
                          max_new_tokens=n_new, eos_token_id=cfg.eot, pad_token_id=cfg.eot, bos_token_id=cfg.eot,
                          decoder_start_token_id=cfg.sot, return_dict_in_generate=True, output_scores=True)
    enc = oracle.encode(feats)
    got = oracle.generate(enc, [prompt] * B, beam_size=K, patience=1.0, num_hypotheses=K, length_penalty=0.0,
                          max_length=len(prompt) + n_new, suppress_tokens=sup, suppress_blank=True)
    early = 0
    with torch.no_grad():
        for b in range(B):
            out = GenerationMixin.generate(hf, input_features=torch.from_numpy(feats[b:b + 1]),
                                           decoder_input_ids=torch.tensor([prompt]), generation_config=gc,
                                           logits_processor=procs)
            seqs = out.sequences[:, len(prompt):].tolist()
            hf_hyps = []
            for s in seqs:
                s = s[:s.index(cfg.eot)] if cfg.eot in s else s       # (padding after <eot> is <eot>)
                hf_hyps.append(s)
            hf_scores = out.sequences_scores.tolist()
            o_hyps, o_scores = got[b].sequences_ids, got[b].scores
            early += sum(len(h) < n_new for h in o_hyps)
            print(f"seed {seed} lift {eot_lift} chunk {b}: lengths oracle {[len(h) for h in o_hyps]} hf {[len(h) for h in hf_hyps]}, "
                  f"best score {o_scores[0]:.5f} / {hf_scores[0]:.5f}")
            assert o_hyps[0] == hf_hyps[0], (b, o_hyps[0], hf_hyps[0])
            assert abs(o_scores[0] - hf_scores[0]) < 2e-3 * max(1.0, abs(hf_scores[0]))
            # every returned hypothesis, in order (scores are distinct on random weights)
            assert o_hyps == hf_hyps, (b, o_hyps, hf_hyps)
            assert np.abs(np.asarray(o_scores) - np.asarray(hf_scores)).max() < 2e-3 * max(1.0, abs(hf_scores[-1]))
    if eot_lift >= 2.5:
        assert early > 0, "the lifted <eot> logit did not finish any hypothesis before the budget: the case tests nothing new"


def test_detect_language_and_align_match_a_pipeline_built_from_transformers_parts():
    """oracle.detect_language and oracle.align against the same quantities assembled from independent parts: transformers'
    Whisper forward (logits, and the cross-attention probabilities it returns with output_attentions), transformers' ports of
    openai-whisper's median filter and DTW.  The row selection (<|notimestamps|> .. last text token) and the crop at
    num_frames // 2 are openai-whisper timing.py's / the reference's (transcribe.py:1709-1746) and are restated here."""
    gw = pytest.importorskip("transformers.models.whisper.generation_whisper")
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=31, dtype=np.float32)
    hf = _load_into_hf(cfg, w)
    try:
        hf.config._attn_implementation = "eager"       # (cross-attention probabilities are only returned by the eager path)
        hf.model.config._attn_implementation = "eager"
    except Exception:
        pass
    oracle = OracleWhisper(cfg, w, emulate_fp16=False)
    rng = np.random.default_rng(31)
    B = 2
    feats = rng.standard_normal((B, cfg.n_mels, 3000)).astype(np.float32) * 0.5
    enc = oracle.encode(feats)
    # ---- detect_language: one step on <sot>, softmax over the language ids only
    got = oracle.detect_language(enc)
    with torch.no_grad():
        lg = hf(input_features=torch.from_numpy(feats), decoder_input_ids=torch.full((B, 1), cfg.sot)).logits[:, 0]
    pr = torch.softmax(lg[:, cfg.lang_begin:cfg.lang_begin + cfg.n_langs].float(), -1).numpy()
    for b in range(B):
        ids = [t for t, _ in got[b]]
        assert ids == [cfg.lang_begin + int(i) for i in np.argsort(-pr[b], kind="stable")]
        assert np.abs(np.asarray([p for _, p in got[b]]) - np.sort(pr[b])[::-1]).max() < 1e-5
    # ---- align
    start = list(cfg.sot_sequence)
    texts = [[int(t) for t in rng.integers(20, 300, size=9)], [int(t) for t in rng.integers(20, 300, size=5)]]
    num_frames = [3000, 1830]
    width = 7
    res = oracle.align(enc, start, texts, num_frames, median_filter_width=width)
    heads = oracle.alignment_heads()
    for b in range(B):
        toks = start + [cfg.no_timestamps] + texts[b] + [cfg.eot]
        with torch.no_grad():
            out = hf(input_features=torch.from_numpy(feats[b:b + 1]), decoder_input_ids=torch.tensor([toks]),
                     output_attentions=True)
        assert out.cross_attentions is not None and out.cross_attentions[0] is not None
        n0 = len(start) + 1
        tp = torch.softmax(out.logits[0, n0 - 1:n0 - 1 + len(texts[b])].float(), -1)
        want_probs = [float(tp[i, t]) for i, t in enumerate(texts[b])]
        assert np.abs(np.asarray(res[b].text_token_probs) - np.asarray(want_probs)).max() < 1e-5
        nfr = min(cfg.n_audio_ctx, num_frames[b] // 2)
        wts = torch.stack([out.cross_attentions[l][0, h] for (l, h) in heads])[:, :, :nfr].float()
        wts = (wts - wts.mean(dim=-2, keepdim=True)) / wts.std(dim=-2, keepdim=True, unbiased=False)
        wts = gw._median_filter(wts, width)
        m = wts.mean(dim=0)[n0 - 1:-1]
        ti, fi = gw._dynamic_time_warping(-m.double().numpy())
        assert res[b].alignments == list(zip(ti.tolist(), fi.tolist())), b
