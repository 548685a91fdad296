"""End-to-end parity of the HIP engine (through the ctranslate2-compatible front and the C ABI)
against the CPU oracle on seeded synthetic weights: encode, generate (greedy + beam),
detect_language, align.

The oracle is run with fp16 rounding emulated at the engine's storage points, so what is left
is accumulation-order noise.  Token ids must match bit-exactly wherever the oracle's own
top-1/top-2 margin exceeds MARGIN (a flip at a smaller margin is fp noise, not a bug);
scores / probabilities within 1e-3 (north_star tolerance)."""
import numpy as np
import pytest

from conftest import greedy_gaps, bench_audio, check_hypothesis, forced_result, forced_score, make_model

pytestmark = pytest.mark.gpu

MARGIN = 2e-2


@pytest.fixture(scope="module", params=["micro", "tiny.en"])
def setup(request):
    from oracle.whisper import OracleWhisper
    cfg, w, model = make_model(request.param, seed=11, max_batch=4, max_beam=5)
    oracle = OracleWhisper(cfg, w, emulate_fp16=True)
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


def test_encode(setup):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    assert enc.shape == [3, 1500, cfg.d_model]
    got = enc.to_numpy()
    ref = oracle.encode(feats)
    err = float(np.abs(got - ref).max())
    rel = err / float(np.abs(ref).max())
    print(f"[{cfg.name}] encoder: max abs err {err:.3e} (rel {rel:.2e}), ref absmax {np.abs(ref).max():.2f}")
    assert rel < 1e-2
    # fused resident path == host-features path
    chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
    enc2 = model.encode_pcm(chunks).to_numpy()
    err2 = float(np.abs(enc2 - got).max())
    print(f"[{cfg.name}] encode_pcm vs encode(features): {err2:.3e}")
    assert err2 < 2e-2 * max(1.0, float(np.abs(ref).max()))


def _check_ids(got, ref):
    """ids must agree up to the first step where the oracle's margin is below MARGIN"""
    n = 0
    while n < len(ref.sequences_ids[0]) and n < len(ref.margins) and ref.margins[n] > MARGIN:
        n += 1
    assert got.sequences_ids[0][:n] == ref.sequences_ids[0][:n], (got.sequences_ids[0], ref.sequences_ids[0], ref.margins)
    return n, got.sequences_ids[0] == ref.sequences_ids[0]


@pytest.mark.parametrize("timestamps", [False, True])
def test_generate_greedy(setup, timestamps):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()   # feed the oracle the engine's own encoder output: isolates the decoder
    prompt = _prompt(cfg, timestamps)
    L = 24
    kw = dict(beam_size=1, max_length=len(prompt) + L, suppress_blank=True, suppress_tokens=_suppress(cfg),
              max_initial_timestamp_index=50)
    got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, **kw)
    for b, (g, r) in enumerate(zip(got, ref)):
        n, same = _check_ids(g, r)          # ids identical wherever the oracle's margin is above MARGIN
        # the score is checked ALWAYS: against the oracle's score of the engine's own ids (equal to the oracle's
        # own score when the ids agree)
        sf = forced_score(oracle, enc_np[b], prompt, g.sequences_ids[0], kw)
        print(f"[{cfg.name}] greedy ts={timestamps} chunk {b}: {n}/{len(r.sequences_ids[0])} margin-safe ids equal, "
              f"all equal={same}, score {g.scores[0]:.5f} vs {sf:.5f} (oracle's own path {r.scores[0]:.5f}), "
              f"no_speech {g.no_speech_prob:.3e} vs {r.no_speech_prob:.3e}")
        assert abs(g.scores[0] - sf) < 1e-3 * max(1.0, abs(sf))
        assert abs(g.no_speech_prob - r.no_speech_prob) < 1e-3
        # the id prefix above ends at the first numerically tied step (it may be step 1); this does not: EVERY token the
        # engine chose is the oracle's arg-max given the engine's own history, or within the noise margin of it
        gaps = greedy_gaps(oracle, enc_np[b], prompt, g.sequences_ids[0], kw)
        assert len(gaps) >= len(g.sequences_ids[0]) >= 4 and max(gaps) <= MARGIN, (b, max(gaps), gaps)
        print(f"[{cfg.name}]   every one of the {len(gaps)} choices within {max(gaps):.2e} of the oracle's arg-max")


def test_generate_teacher_forced_logprobs(setup):
    """per-step parity independent of argmax ties: force the oracle along the engine's ids and compare
    the cumulative log-prob"""
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg)
    L = 16
    kw = dict(beam_size=1, max_length=len(prompt) + L, suppress_tokens=_suppress(cfg), length_penalty=0.0)
    got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, force_tokens=[g.sequences_ids[0] for g in got], **kw)
    for g, r in zip(got, ref):
        assert r.sequences_ids[0] == g.sequences_ids[0]
        print(f"[{cfg.name}] teacher-forced cum logprob {g.scores[0]:.5f} vs {r.scores[0]:.5f}")
        assert abs(g.scores[0] - r.scores[0]) < 2e-3 * max(1.0, abs(r.scores[0]))


@pytest.mark.parametrize("beam,timestamps", [(5, False), (5, True), (2, False)])
def test_generate_beam(setup, beam, timestamps):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg, timestamps)
    L = 16
    kw = dict(beam_size=beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L,
              suppress_tokens=_suppress(cfg))
    got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, **kw)
    for b, (g, r) in enumerate(zip(got, ref)):
        check_hypothesis(oracle, enc_np[b], prompt, g, r, kw, what=f"[{cfg.name}] beam={beam} ts={timestamps} chunk {b}")
        assert abs(g.no_speech_prob - r.no_speech_prob) < 1e-3


def test_generate_eot_and_early_finish(setup):
    """make <eot> very likely after a few tokens via min_new_tokens=0 and a tiny budget: exercises
    finished-hypothesis bookkeeping and the max-length finalisation"""
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg)
    for beam in (1, 3):
        kw = dict(beam_size=beam, max_length=len(prompt) + 3, suppress_tokens=None, suppress_blank=False)
        got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
        ref = oracle.generate(enc_np, [prompt] * 3, **kw)
        for b, (g, r) in enumerate(zip(got, ref)):
            assert len(g.sequences_ids[0]) <= 3
            check_hypothesis(oracle, enc_np[b], prompt, g, r, kw, search=beam > 1,
                             what=f"[{cfg.name}] budget-3 beam={beam} chunk {b}")
    # <eot> allowed from the first step and made attractive by suppressing (almost) everything else: hypotheses
    # finish early, the finished list fills up and the run stops before the budget
    keep = {cfg.eot, 20, 21, 22}
    sup = [t for t in range(cfg.n_vocab) if t not in keep]
    for beam in (1, 4):
        kw = dict(beam_size=beam, max_length=len(prompt) + 6, suppress_tokens=sup, suppress_blank=False)
        got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
        ref = oracle.generate(enc_np, [prompt] * 3, **kw)
        for b, (g, r) in enumerate(zip(got, ref)):
            assert all(t in keep for t in g.sequences_ids[0])
            check_hypothesis(oracle, enc_np[b], prompt, g, r, kw, search=beam > 1,
                             what=f"[{cfg.name}] eot-heavy beam={beam} chunk {b}")


def test_generate_sampling(setup):
    """temperature fallback of the sequential path (transcribe.py:1433-1439): beam_size=1, sampling_topk=0,
    num_hypotheses=best_of.  The engine and the oracle share the counter-based Gumbel hash, so the same
    seed must give the same samples (up to float noise at near-ties of the perturbed keys)."""
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    prompt = _prompt(cfg)
    kw = dict(beam_size=1, num_hypotheses=5, sampling_topk=0, sampling_temperature=0.8, max_length=len(prompt) + 12,
              suppress_tokens=_suppress(cfg), seed=1234)
    got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, **kw)
    same = total = 0
    for g, r in zip(got, ref):
        assert len(g.sequences_ids) == 5 and all(len(s) == 12 for s in g.sequences_ids)
        assert g.scores == sorted(g.scores, reverse=True)
        assert len({tuple(s) for s in g.sequences_ids}) >= 4          # samples differ from each other
        gs = {tuple(s): sc for s, sc in zip(g.sequences_ids, g.scores)}
        for s, sc in zip(r.sequences_ids, r.scores):
            total += 1
            if tuple(s) in gs:
                same += 1
                assert abs(gs[tuple(s)] - sc) < 2e-3 * max(1.0, abs(sc))
    print(f"[{cfg.name}] sampling: {same}/{total} sampled sequences identical to the oracle's")
    assert same >= int(0.8 * total)
    # a different seed gives different samples; greedy is unaffected by the seed
    other = model.generate(enc, [prompt] * 3, **dict(kw, seed=99))
    assert other[0].sequences_ids != got[0].sequences_ids
    with pytest.raises(ValueError):
        model.generate(enc, [prompt] * 3, beam_size=1, sampling_topk=7)


def test_generate_argument_errors(setup):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    with pytest.raises(ValueError):
        model.generate(enc, [[cfg.sot]] * 2)                      # wrong number of prompts
    with pytest.raises(ValueError):
        model.generate(enc, [[cfg.sot], [cfg.sot, cfg.sot], [cfg.sot]])  # ragged prompts
    with pytest.raises(ValueError):
        model.generate(enc, [[cfg.sot]] * 3, beam_size=99)
    with pytest.raises(ValueError):
        model.encode(StorageView.from_array(np.zeros((1, cfg.n_mels, 100), np.float32)))


def test_detect_language(setup):
    from faster_whisper_amd.backend import StorageView, language_token_strings
    cfg, model, oracle, feats = setup
    if not cfg.is_multilingual:
        with pytest.raises(RuntimeError):
            model.detect_language(StorageView.from_array(feats))
        return
    enc = model.encode(StorageView.from_array(feats))
    got = model.detect_language(enc)
    ref = oracle.detect_language(enc.to_numpy())
    names = language_token_strings(cfg)
    for g, r in zip(got, ref):
        assert len(g) == cfg.n_langs
        assert abs(sum(p for _, p in g) - 1.0) < 1e-4
        gp = {tok: p for tok, p in g}
        for tid, p in r:
            assert abs(gp[names[tid - cfg.lang_begin]] - p) < 1e-3
        print(f"[{cfg.name}] detect_language top: {g[0]} vs {(names[r[0][0] - cfg.lang_begin], r[0][1])}")


def test_align(setup):
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    enc = model.encode(StorageView.from_array(feats))
    enc_np = enc.to_numpy()
    rng = np.random.default_rng(4)
    text = [rng.integers(10, 300, size=n).tolist() for n in (12, 5, 20)]
    num_frames = [3000, 1250, 2000]
    got = model.align(enc, cfg.sot_sequence, text, num_frames, median_filter_width=7)
    ref = oracle.align(enc_np, cfg.sot_sequence, text, num_frames, median_filter_width=7)
    for b, (g, r) in enumerate(zip(got, ref)):
        pe = float(np.abs(np.array(g.text_token_probs) - np.array(r.text_token_probs)).max())
        # DTW paths are discrete: compare the word-boundary times the reference derives from them
        gt = np.array([t for _, t in g.alignments])
        rt = np.array([t for _, t in r.alignments])
        gi = np.array([i for i, _ in g.alignments])
        ri = np.array([i for i, _ in r.alignments])
        gj = gt[np.r_[True, np.diff(gi) > 0]]
        rj = rt[np.r_[True, np.diff(ri) > 0]]
        # n_text + 1 rows: the <|notimestamps|> position (predicting the first text token) .. the last text token
        assert len(gj) == len(rj) == len(text[b]) + 1
        jd = int(np.abs(gj - rj).max())
        print(f"[{cfg.name}] align chunk {b}: token prob err {pe:.2e}, max jump-time diff {jd} frames, "
              f"path len {len(g.alignments)} vs {len(r.alignments)}")
        assert pe < 1e-3
        assert gi[0] == 0 and gi[-1] == len(text[b]) and gt[-1] == num_frames[b] // 2 - 1
        assert jd <= 2


def test_fp32_reference_end_to_end(setup):
    """BASELINE.json: "outputs match the reference CTranslate2 CPU path" — fp32 arithmetic.  Every other test of this file
    compares the engine with the oracle that rounds to fp16 where the engine does and decodes from the engine's own
    encoder output (what is left is accumulation order).  Here the oracle runs WITHOUT fp16 emulation (the fp16-stored
    weights, every activation in fp32) and END TO END from the PCM: its own log-mel (oracle/logmel.py = the reference's
    feature extractor), its own encoder output; the engine's greedy ids / beam-5 hypothesis are scored by teacher forcing.
    The north-star tolerance (1e-3 on log-probs and probabilities) is asserted on every quantity."""
    from oracle.logmel import log_mel_chunks
    from oracle.whisper import OracleWhisper
    from faster_whisper_amd.backend import language_token_strings
    cfg, model, oracle, feats = setup
    o32 = OracleWhisper(cfg, oracle_weights(oracle), emulate_fp16=False)
    chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
    enc = model.encode_pcm(chunks)
    got = enc.to_numpy()
    enc32 = o32.encode(log_mel_chunks(chunks, cfg.n_mels))
    rel = float(np.abs(got - enc32).max() / np.abs(enc32).max())
    print(f"[{cfg.name} vs fp32] encoder output, end to end from the PCM: max rel {rel:.2e}")
    assert rel < 1e-2
    prompt = _prompt(cfg)
    L = 24
    kw1 = dict(beam_size=1, max_length=len(prompt) + L, suppress_tokens=_suppress(cfg), length_penalty=0.0)
    kw5 = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, suppress_tokens=_suppress(cfg))
    g1 = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw1)
    g5 = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw5)
    for b in range(3):
        r1 = forced_result(o32, enc32[b], prompt, g1[b].sequences_ids[0], kw1)
        r5 = forced_result(o32, enc32[b], prompt, g5[b].sequences_ids[0], kw5)
        n1 = len(g1[b].sequences_ids[0])
        d1 = abs(g1[b].scores[0] - r1.scores[0]) / max(1, n1)
        d5 = abs(g5[b].scores[0] - r5.scores[0]) / max(1.0, abs(r5.scores[0]))
        dn = abs(g5[b].no_speech_prob - r5.no_speech_prob)
        gaps = np.array(r1.forced_gaps)
        print(f"[{cfg.name} vs fp32] chunk {b}: greedy {n1} steps, per token {d1:.2e}; beam-5 score {g5[b].scores[0]:.5f} vs "
              f"{r5.scores[0]:.5f} (rel {d5:.2e}); no_speech diff {dn:.1e}; {int((gaps == 0).sum())}/{len(gaps)} greedy ids are "
              f"the fp32 arg-max (largest gap {gaps.max():.4f})")
        assert d1 < 1e-3 and d5 < 1e-3 and dn < 1e-3, (b, d1, d5, dn)
        assert gaps.max() <= MARGIN, (b, gaps)
    if cfg.is_multilingual:
        names = language_token_strings(cfg)
        for b, (g, r) in enumerate(zip(model.detect_language(enc), o32.detect_language(enc32))):
            gp = dict(g)
            worst = max(abs(gp[names[tid - cfg.lang_begin]] - p) for tid, p in r)
            print(f"[{cfg.name} vs fp32] chunk {b}: language probabilities max diff {worst:.2e}")
            assert worst < 1e-3
    rng = np.random.default_rng(4)
    text = [rng.integers(10, 300, size=n).tolist() for n in (12, 5, 20)]
    num_frames = [3000, 1250, 2000]
    ga = model.align(enc, cfg.sot_sequence, text, num_frames, median_filter_width=7)
    ra = o32.align(enc32, cfg.sot_sequence, text, num_frames, median_filter_width=7)
    for b, (g, r) in enumerate(zip(ga, ra)):
        pe = float(np.abs(np.array(g.text_token_probs) - np.array(r.text_token_probs)).max())
        gi, gt = np.array([i for i, _ in g.alignments]), np.array([t for _, t in g.alignments])
        ri, rt = np.array([i for i, _ in r.alignments]), np.array([t for _, t in r.alignments])
        jd = int(np.abs(gt[np.r_[True, np.diff(gi) > 0]] - rt[np.r_[True, np.diff(ri) > 0]]).max())
        print(f"[{cfg.name} vs fp32] align chunk {b}: token prob err {pe:.2e}, word-boundary diff {jd} frames")
        assert pe < 1e-3 and jd <= 2


def oracle_weights(oracle):
    """the weight dict an OracleWhisper was built from (fp16-representable values: rounding them again is the identity)"""
    return {k: v.numpy() for k, v in oracle.w.items()}


@pytest.mark.parametrize("name", ["micro", "tiny.en"])
def test_generate_greedy_literal_ids_peaked(name):
    """north star: "token ids bit-exact at beam_size=1 greedy" — literally, on the PEAKED variant of the synthetic
    weights (weights.make_peaked; SURVEY.md section 7: random weights tie at the noise level at some step of every
    transcript, a trained model does not): 48 free-running greedy steps on 3 chunks, with and without timestamps; the
    ids must equal the oracle's own free-running greedy ids, all of them, and the oracle's top-1 / top-2 margin must
    clear the fp16 noise MARGIN at every step so that the equality is forced, not lucky."""
    from faster_whisper_amd import get_config, synthetic_weights
    from faster_whisper_amd.backend import StorageView
    from oracle.whisper import OracleWhisper
    cfg = get_config(name)
    w = synthetic_weights(cfg, seed=7, peaked=True)
    _, _, model = make_model(name, max_batch=4, max_beam=5, cfg=cfg, weights=w)
    oracle = OracleWhisper(cfg, w, emulate_fp16=True)
    chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
    enc = model.encode(StorageView.from_array(model.log_mel(chunks)))
    enc_np = enc.to_numpy()
    L = 48
    for timestamps in (False, True):
        prompt = _prompt(cfg, timestamps)
        kw = dict(beam_size=1, max_length=len(prompt) + L, suppress_blank=True, suppress_tokens=_suppress(cfg),
                  max_initial_timestamp_index=50)
        got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
        ref = oracle.generate(enc_np, [prompt] * 3, **kw)
        for b, (g, r) in enumerate(zip(got, ref)):
            m = np.array(r.margins)
            print(f"[{cfg.name} peaked] ts={timestamps} chunk {b}: {len(r.sequences_ids[0])} oracle ids, margins min "
                  f"{m.min():.3f} median {np.median(m):.2f}; engine ids equal: {g.sequences_ids[0] == r.sequences_ids[0]}")
            assert len(r.sequences_ids[0]) >= 8
            if not timestamps:
                # (with timestamps the rules force ts tokens whose margin is set by the rule, not by the peaks)
                assert m.min() > 2 * MARGIN, ("not peaked enough", m.min())
            safe = m.min() > 2 * MARGIN
            if safe:
                assert g.sequences_ids[0] == r.sequences_ids[0], (b, g.sequences_ids[0], r.sequences_ids[0])
                assert abs(g.scores[0] - r.scores[0]) < 1e-3 * max(1.0, abs(r.scores[0]))
            else:
                n, _ = _check_ids(g, r)
                assert n >= 1
        if not timestamps:
            # the same claim against the fp32 oracle running END TO END and free (its own log-mel, encoder, greedy path):
            # the reference's CPU path is fp32 arithmetic — literal id equality, all 48 steps
            from oracle.logmel import log_mel_chunks
            o32 = OracleWhisper(cfg, w, emulate_fp16=False)
            r32 = o32.generate(o32.encode(log_mel_chunks(chunks, cfg.n_mels)), [prompt] * 3, **kw)
            for b, (g, r) in enumerate(zip(got, r32)):
                m = np.array(r.margins)
                print(f"[{cfg.name} peaked vs fp32, end to end] chunk {b}: fp32 margins min {m.min():.3f}; engine ids equal: "
                      f"{g.sequences_ids[0] == r.sequences_ids[0]}; score {g.scores[0]:.5f} vs {r.scores[0]:.5f}")
                assert m.min() > 2 * MARGIN
                assert g.sequences_ids[0] == r.sequences_ids[0], (b, g.sequences_ids[0], r.sequences_ids[0])
                assert abs(g.scores[0] - r.scores[0]) < 1e-3 * max(1.0, abs(r.scores[0]))


def test_self_attention_forms_return_the_same_bits(setup):
    """Round 5: the decoder self-attention has a latency form (solo runs) and a throughput form (merged runs) next to the
    first form of rounds 1-4 (dec_kernels.hip: dec_self_attn2_kernel).  All of them keep the row's arithmetic — same dot
    products, same reductions, the same ascending-position accumulation — so which one a launch takes must not change a
    bit of any result: beam search (ids, scores, no-speech) over 40 steps (the slot table is exercised by the beam
    reorders, 40 + prompt positions span two 32-position batches of the throughput form) and greedy with timestamps."""
    from faster_whisper_amd import _lib
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    lib = _lib.load()
    enc = model.encode(StorageView.from_array(feats))
    res = {}
    try:
        for form in (1, 2, 3, 0):
            _lib.check(lib.fw_test_knob(2, form))
            out = []
            for beam, ts in ((5, False), (1, True), (3, True)):
                prompt = _prompt(cfg, ts)
                kw = dict(beam_size=beam, max_length=len(prompt) + 40, suppress_blank=True, suppress_tokens=_suppress(cfg),
                          max_initial_timestamp_index=50, num_hypotheses=min(beam, 2))
                got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
                out.append([(g.sequences_ids, [np.float32(s).tobytes() for s in g.scores],
                             np.float32(g.no_speech_prob).tobytes()) for g in got])
            res[form] = out
    finally:
        _lib.check(lib.fw_test_knob(2, 0))
    for form in (2, 3, 0):
        assert res[form] == res[1], f"self-attention form {form} differs from the first form"
    print(f"[{cfg.name}] self-attention forms 1 / 2 / 3 / auto: identical ids, scores and no-speech bits over 3 configurations")


def test_encoder_vt_epilogue_same_bits(setup):
    """Round 5 (fw_test_knob 5, default on): the encoder's V^T projection stores its transposed tile through LDS in whole
    row segments of Ct [d][t_pad] (M = 1 500 keys, row stride 1 536: the last 16-byte chunk of a row is partial, the padding
    columns are never written) instead of 32 scattered 8-byte stores per lane.  Same arithmetic per element: the encoder
    output — every layer's attention reads V^T — may not move by a bit, ragged and empty chunks included."""
    from faster_whisper_amd import _lib
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    lib = _lib.load()
    out = {}
    try:
        for on in (0, 1):
            _lib.check(lib.fw_test_knob(5, on))
            out[on] = model.encode(StorageView.from_array(feats)).to_numpy()
    finally:
        _lib.check(lib.fw_test_knob(5, 1))
    assert np.isfinite(out[1]).all() and np.array_equal(out[0], out[1])
    print(f"[{cfg.name}] encoder output with the staged / direct V^T epilogue: bit-identical")


def test_cross_kv_layered_same_bits(setup):
    """Round 5 (fw_test_knob 6, default on): the cross-attention K / V^T projections of ALL decoder layers run as two
    launches of the encoder GEMM (its n tiles run over the layers' weights) instead of two per layer.  Same tiles, same
    arithmetic per element, same destinations in the pool: generate (beam and greedy), detect_language and align — every
    consumer of the pool — return the same bits.  Each setting gets a FRESH encoder output, so that the pool cannot hand
    back the block the other setting filled."""
    from faster_whisper_amd import _lib
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    lib = _lib.load()
    rng = np.random.default_rng(45)
    text = [rng.integers(10, 300, size=n).tolist() for n in (9, 5, 12)]
    res = {}
    try:
        for on in (0, 1):
            _lib.check(lib.fw_test_knob(6, on))
            enc = model.encode(StorageView.from_array(feats))
            out = []
            for beam in (5, 1):
                prompt = _prompt(cfg, True)
                kw = dict(beam_size=beam, max_length=len(prompt) + 12, suppress_blank=True, suppress_tokens=_suppress(cfg))
                got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
                out.append([(g.sequences_ids, [np.float32(s).tobytes() for s in g.scores],
                             np.float32(g.no_speech_prob).tobytes()) for g in got])
            enc2 = model.encode(StorageView.from_array(feats))          # align fills a block of its own
            al = model.align(enc2, cfg.sot_sequence, text, [3000, 1250, 2000], median_filter_width=7)
            out.append([(a.alignments, np.asarray(a.text_token_probs, np.float32).tobytes()) for a in al])
            if cfg.is_multilingual:
                enc3 = model.encode(StorageView.from_array(feats))
                out.append(model.detect_language(enc3))
            res[on] = out
    finally:
        _lib.check(lib.fw_test_knob(6, 1))
    for i, (a, b) in enumerate(zip(res[0], res[1])):
        assert a == b, f"layered cross-K/V projection changed result set {i}"
    print(f"[{cfg.name}] cross-K/V projections per layer / layered: identical ids, scores, no-speech, alignments, languages")


def test_position_blocks_same_bits(setup):
    """Round 5 (fw_test_knob 4, default on): the prompt forward of `generate` and the teacher-forced pass of `align` go
    through the decoder in blocks of up to 16 positions per pass (rows = chunks x positions; a sibling position's K / V
    are read from the qkv buffer instead of the cache) instead of one position per pass.  Every row keeps its arithmetic,
    so nothing may move by a bit: a 4-token prompt (one block of 3), a prompt with 37 tokens of previous text (blocks of
    16 + 16 + 4 with the <sot> row — the no-speech probability — inside the third), beam and greedy; align over texts
    of 5 .. 40 tokens (the engine pads to the longest: several blocks, ragged ends)."""
    from faster_whisper_amd import _lib
    from faster_whisper_amd.backend import StorageView
    cfg, model, oracle, feats = setup
    lib = _lib.load()
    enc = model.encode(StorageView.from_array(feats))
    rng = np.random.default_rng(44)
    prev = [cfg.sot_prev] + rng.integers(10, 300, size=36).tolist()
    text = [rng.integers(10, 300, size=n).tolist() for n in (33, 5, 40)]
    num_frames = [3000, 1250, 2000]
    res = {}
    try:
        for on in (0, 1):
            _lib.check(lib.fw_test_knob(4, on))
            out = []
            for prompt, beam in ((_prompt(cfg, True), 5), (prev + _prompt(cfg, False), 5), (prev + _prompt(cfg, True), 1)):
                kw = dict(beam_size=beam, max_length=len(prompt) + 12, suppress_blank=True, suppress_tokens=_suppress(cfg),
                          max_initial_timestamp_index=50)
                got = model.generate(enc, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
                out.append([(g.sequences_ids, [np.float32(s).tobytes() for s in g.scores],
                             np.float32(g.no_speech_prob).tobytes()) for g in got])
            al = model.align(enc, cfg.sot_sequence, text, num_frames, median_filter_width=7)
            out.append([(a.alignments, np.asarray(a.text_token_probs, np.float32).tobytes()) for a in al])
            res[on] = out
    finally:
        _lib.check(lib.fw_test_knob(4, 1))
    for i, (a, b) in enumerate(zip(res[0], res[1])):
        assert a == b, f"position blocks changed result set {i}"
    print(f"[{cfg.name}] position blocks on / off: identical ids, scores, no-speech, alignments and token probabilities")
