"""The oracle's `fold_ln` evaluation order (decoder LayerNorms folded into the consuming linears, as the engine
evaluates them) is the same function as the explicit order: without fp16 rounding the two agree to float32
round-off on every decoder entry point (generate scores, language probabilities, alignment probabilities)."""
import numpy as np

from conftest import bench_audio


def test_fold_order_is_the_same_function_in_fp32():
    from faster_whisper_amd import get_config, synthetic_weights
    from oracle import logmel as olm
    from oracle.whisper import OracleWhisper
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=7)
    o = OracleWhisper(cfg, w, emulate_fp16=False)
    feats = olm.log_mel_chunks([bench_audio(480000, seed=3), bench_audio(300000, seed=4)], cfg.n_mels)
    enc = o.encode(feats)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=3, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 6)

    def run():
        g = o.generate(enc, [prompt] * 2, **kw)
        lang = o.detect_language(enc) if cfg.is_multilingual else None
        text = [[t for t in r.sequences_ids[0] if t < cfg.eot] for r in g]
        al = o.align(enc, cfg.sot_sequence, text, [3000, 1875], median_filter_width=7)
        return g, lang, al
    g0, l0, a0 = run()
    o.fold_ln = True
    g1, l1, a1 = run()
    assert o._folded, "the folded path did not run"
    for x, y in zip(g0, g1):
        assert x.sequences_ids == y.sequences_ids
        assert abs(x.scores[0] - y.scores[0]) < 2e-5
    if l0 is not None:
        for x, y in zip(l0, l1):
            dx, dy = dict(x), dict(y)
            assert max(abs(dx[k] - dy[k]) for k in dx) < 2e-5
    for x, y in zip(a0, a1):
        assert x.alignments == y.alignments
        assert np.abs(np.array(x.text_token_probs) - np.array(y.text_token_probs)).max() < 2e-5
