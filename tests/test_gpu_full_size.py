"""Parity at BASELINE.json's benchmarked geometry: large-v3 shapes (d = 1280, ffn 5120, 32 + 32 layers, 128 mels,
vocabulary 51 866), **16 chunks x beam 5 = 80 decoder rows** — the configuration bench.py times — on seeded
synthetic weights, for float16, int8_float16 (C3) and distil-large-v3 (C5: 2 decoder layers).

The engine always runs the whole batch of 16; the CPU oracle (fp16 storage emulated; int8 restated exactly) is run
on a 2-chunk subset (first chunk, and one in the last row tile) because a large-v3 beam step costs about a second
of CPU.  What is compared, per configuration:
  * encoder output of one chunk (relative max / rms error);
  * >= 8 teacher-forced greedy steps: log-prob of the engine's own ids under the oracle, per token;
  * beam 5: score of the engine's hypothesis under the oracle, ids identical or tied (conftest.check_hypothesis),
    no-speech probability;
  * detect_language probabilities; align: token probabilities, word-boundary frames <= 2.
Tolerances.  At this depth (32 + 32 layers of random weights) ANY two fp16 evaluations of the model differ at the
1e-3 level, on the CPU alone (tests/numerics_ln_fold_noise.py -> tests/golden/numerics_ln_fold_noise.txt): fp16
storage vs fp32 up to 8.5e-4 per token and 3.0e-3 on a language probability; the engine's evaluation order (decoder
LayerNorms folded into the linears they feed, oracle `fold_ln`) vs the explicit order up to 1.4e-3 per token, 2.0e-3
on a beam-5 score, 2.9e-3 on a language probability; two implementations of the SAME folded order still 6e-4 per
token.  The north-star 1e-3 is therefore asserted where the model is shallow enough for it to be meaningful (micro,
tiny.en: tests/test_gpu_model.py; distil-large-v3's 2-layer decoder here), every kernel is checked alone at these
shapes at fp16 round-off (tests/test_gpu_kernels.py), and the large-v3 end-to-end figures are held to 2x the
largest CPU-observed difference between valid fp16 orders: 3e-3 per token, 4e-3 beam score, 6e-3 / 5e-3 language /
token probabilities.  A wiring error (wrong weight, wrong row, wrong layer) moves these by 1e-1, not 1e-3.
(The int8_float16 tolerances and where they come from are next to the tolerance table in _run.)
Reference call sites: transcribe.py:222-246 (generate + score), :1709-1746 (align), :1823-1828 (detect_language)."""
import os

import numpy as np
import pytest

from conftest import bench_audio, check_hypothesis, forced_score

pytestmark = pytest.mark.gpu

B = 16
SUBSET = (0, 13)      # chunks the oracle is run on: rows 0-4 (first tile) and 65-69 (last tile of the 80)


@pytest.fixture(scope="module")
def lv3():
    import torch
    from faster_whisper_amd import get_config, synthetic_weights
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_config("large-v3")
    return cfg, synthetic_weights(cfg, seed=1234)


def _chunks():
    out = [bench_audio(480000, seed=100 + i) for i in range(B)]
    out[5] = out[5][:200000]          # ragged: a short chunk and an empty one ride along
    out[11] = out[11][:0]
    return out


def _run(cfg, w, compute_type, tf_steps=8, beam_steps=6):
    from faster_whisper_amd import Whisper
    from faster_whisper_amd.backend import StorageView, language_token_strings
    from oracle.whisper import OracleWhisper
    i8 = compute_type == "int8_float16"
    # Tolerances: module docstring (fp16).  distil-large-v3 (2 decoder layers) keeps the north-star figures.
    # int8_float16: the engine and the oracle quantise activations that differ by fp16 rounding, a flipped int8 code
    # is 1/127 of a row's range and 2 x 32 quantised blocks accumulate them: measured 1.3e-2 rms on the encoder
    # output, 1e-2 per token on an 8-step log-prob, 2e-2 on a beam score, 3.8e-2 on a language probability — every
    # single int8 GEMM is bit-exact against the integer reference (tests/test_gpu_int8.py).
    if i8:
        tol = dict(tf=2e-2, beam=4e-2, gap=6e-2, nsp=1e-2, lang=6e-2, align=5e-2, enc=(6e-2, 2e-2))
    elif cfg.n_dec_layers <= 4:
        tol = dict(tf=1e-3, beam=1e-3, gap=2e-2, nsp=1e-3, lang=2e-3, align=2e-3, enc=(3e-2, 5e-3))
    else:
        tol = dict(tf=3e-3, beam=4e-3, gap=2e-2, nsp=1e-3, lang=6e-3, align=5e-3, enc=(3e-2, 5e-3))
    fails = []

    def expect(cond, msg):
        if not cond:
            fails.append(msg)
            print("  MISMATCH:", msg)
    tag = f"[{cfg.name} {compute_type}]"
    model = Whisper(f"synthetic:{cfg.name}", device="cuda", files={"config": cfg, "weights": w},
                    compute_type=compute_type, max_batch_size=B, max_beam_size=5)
    oracle = OracleWhisper(cfg, w, emulate_fp16=True, int8=i8)
    chunks = _chunks()

    # ---- encoder: one chunk against the oracle, the batch against itself ----
    feats0 = model.log_mel(chunks[:1])
    ref0 = oracle.encode(feats0)
    enc = model.encode_pcm(chunks)
    got = enc.to_numpy()
    assert got.shape == (B, 1500, cfg.d_model) and np.isfinite(got).all()
    rel = float(np.abs(got[0] - ref0[0]).max() / np.abs(ref0).max())
    rms = float(np.sqrt(np.mean((got[0] - ref0[0]) ** 2)) / np.sqrt(np.mean(ref0 ** 2)))
    print(f"{tag} encoder chunk 0 of {B}: max rel {rel:.2e}, rms rel {rms:.2e}")
    expect(rel < tol["enc"][0] and rms < tol["enc"][1], f"encoder error {rel:.2e} / {rms:.2e}")
    sub = got[list(SUBSET)]           # the oracle decodes from the engine's own encoder output: isolates the decoder

    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]

    # ---- >= 8 teacher-forced greedy steps on all 16 chunks ----
    kw = dict(beam_size=1, max_length=len(prompt) + tf_steps, length_penalty=0.0, suppress_tokens=sup)
    g1 = model.generate(enc, [prompt] * B, return_scores=True, return_no_speech_prob=True, **kw)
    assert all(len(g.sequences_ids[0]) == tf_steps for g in g1)
    for j, b in enumerate(SUBSET):
        sf = forced_score(oracle, sub[j], prompt, g1[b].sequences_ids[0], kw)
        print(f"{tag} chunk {b}: teacher-forced cum logprob over {tf_steps} steps {g1[b].scores[0]:.5f} vs {sf:.5f}")
        # per generated token: the north-star tolerance is on avg_logprob = cum / (len + 1) (transcribe.py:241-246)
        expect(abs(g1[b].scores[0] - sf) / tf_steps < tol["tf"], f"teacher-forced chunk {b}: {g1[b].scores[0]} vs {sf}")
        if not i8:     # information: the same ids under the engine's own evaluation order (LayerNorms folded)
            oracle.fold_ln = True
            so = forced_score(oracle, sub[j], prompt, g1[b].sequences_ids[0], kw)
            oracle.fold_ln = False
            print(f"{tag} chunk {b}:   per token {abs(g1[b].scores[0] - sf) / tf_steps:.2e}; against the folded order "
                  f"{so:.5f} ({abs(g1[b].scores[0] - so) / tf_steps:.2e} per token)")

    # ---- beam 5 x 16 chunks = 80 rows (the bench geometry) ----
    kw = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + beam_steps, suppress_tokens=sup)
    g5 = model.generate(enc, [prompt] * B, return_scores=True, return_no_speech_prob=True, **kw)
    r5 = oracle.generate(sub, [prompt] * len(SUBSET), **kw)
    for j, b in enumerate(SUBSET):
        try:
            check_hypothesis(oracle, sub[j], prompt, g5[b], r5[j], kw, tol=tol["beam"], gap=tol["gap"],
                             boundary=2 * tol["beam"],
                             what=f"{tag} beam 5 chunk {b}")
        except AssertionError as e:
            expect(False, f"beam chunk {b}: {e}")
        d = abs(g5[b].no_speech_prob - r5[j].no_speech_prob)
        print(f"{tag} chunk {b}: no_speech {g5[b].no_speech_prob:.3e} vs {r5[j].no_speech_prob:.3e}")
        expect(d < tol["nsp"], f"no_speech chunk {b}: diff {d}")
    # the empty chunk and its neighbours decode like any other (no NaN from an all-padding mel)
    assert all(np.isfinite(g.scores[0]) and len(g.sequences_ids[0]) == beam_steps for g in g5)

    # ---- detect_language ----
    names = language_token_strings(cfg)
    gl = model.detect_language(enc)
    rl = oracle.detect_language(sub)
    for j, b in enumerate(SUBSET):
        gp = dict(gl[b])
        worst = max(abs(gp[names[tid - cfg.lang_begin]] - p) for tid, p in rl[j])
        expect(worst < tol["lang"], f"language probabilities chunk {b}: max diff {worst:.2e}")
        print(f"{tag} chunk {b}: detect_language top {gl[b][0]} vs {(names[rl[j][0][0] - cfg.lang_begin], rl[j][0][1])}, "
              f"max prob diff {worst:.2e}")

    # ---- align (word timestamps) on the greedy tokens ----
    text = [[t for t in g.sequences_ids[0] if t < cfg.eot] for g in g1]
    nf = [3000] * B
    nf[5] = 1250
    ga = model.align(enc, cfg.sot_sequence, text, nf, median_filter_width=7)
    ra = oracle.align(sub, cfg.sot_sequence, [text[b] for b in SUBSET], [nf[b] for b in SUBSET], median_filter_width=7)
    for j, b in enumerate(SUBSET):
        pe = float(np.abs(np.array(ga[b].text_token_probs) - np.array(ra[j].text_token_probs)).max())
        gi, gt = np.array([i for i, _ in ga[b].alignments]), np.array([t for _, t in ga[b].alignments])
        ri, rt = np.array([i for i, _ in ra[j].alignments]), np.array([t for _, t in ra[j].alignments])
        gj, rj = gt[np.r_[True, np.diff(gi) > 0]], rt[np.r_[True, np.diff(ri) > 0]]
        assert len(gj) == len(rj) == len(text[b]) + 1
        jd = int(np.abs(gj - rj).max())
        print(f"{tag} chunk {b}: align token prob err {pe:.2e}, max word-boundary diff {jd} frames")
        expect(pe < tol["align"] and jd <= 2, f"align chunk {b}: prob err {pe:.2e}, boundary diff {jd}")
    assert not fails, fails


def test_large_v3_float16(lv3):
    cfg, w = lv3
    _run(cfg, w, "float16")


def test_large_v3_int8_float16(lv3):
    cfg, w = lv3
    _run(cfg, w, "int8_float16", tf_steps=8, beam_steps=4)


def test_distil_large_v3_float16(lv3):
    """C5 geometry: the large-v3 encoder with a 2-layer decoder (synthetic weights are seeded per tensor name, so
    the distil model is the matching subset of the large-v3 set)"""
    from faster_whisper_amd import get_config
    from faster_whisper_amd.weights import weight_shapes
    _, w = lv3
    cfg = get_config("distil-large-v3")
    wd = {k: w[k] for k in weight_shapes(cfg)}
    _run(cfg, wd, "float16", tf_steps=12, beam_steps=10)
