"""fp16 evaluation order of the decoder LayerNorms against the oracle: large-v3, 16 chunks, the
oracle on chunks 0 and 13; FWAMD_LN_UNFOLD = 0 (all folded), 1 (final LayerNorm explicit), 2 (every decoder LayerNorm
explicit).  Per order: per-token teacher-forced log-prob error (8 steps), 48-step beam score error (the oracle scores the
engine's hypothesis), language probability error, align token probability error — and the time of a solo 100-step run.

    python profiles/ln_unfold_probe.py [model]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import bench_audio, forced_score
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    from faster_whisper_amd.backend import language_token_strings
    from oracle.whisper import OracleWhisper
    name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_config(name)
    w = synthetic_weights(cfg, seed=1234)
    oracle = OracleWhisper(cfg, w, emulate_fp16=True)
    B, SUB = 16, (0, 13)
    chunks = [bench_audio(480000, seed=100 + i) for i in range(B)]
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    names = language_token_strings(cfg)
    os.environ["FWAMD_PACK_PLAIN"] = "1"     # the blob carries the plain weight forms too (opt-in since round 5)
    for order in (0, 1, 2):
        os.environ["FWAMD_LN_UNFOLD"] = str(order)
        model = Whisper(f"synthetic:{name}", device="cuda", files={"config": cfg, "weights": w}, compute_type="float16",
                        max_batch_size=B, max_beam_size=5)
        enc = model.encode_pcm(chunks)
        sub = enc.to_numpy()[list(SUB)]
        kw = dict(beam_size=1, max_length=len(prompt) + 8, length_penalty=0.0, suppress_tokens=sup)
        g1 = model.generate(enc, [prompt] * B, return_scores=True, **kw)
        tf = [abs(g1[b].scores[0] - forced_score(oracle, sub[j], prompt, g1[b].sequences_ids[0], kw)) / 8
              for j, b in enumerate(SUB)]
        kw5 = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 48, suppress_tokens=sup)
        g5 = model.generate(enc, [prompt] * B, return_scores=True, **kw5)
        bs = []
        for j, b in enumerate(SUB):
            sf = forced_score(oracle, sub[j], prompt, g5[b].sequences_ids[0], kw5)
            bs.append(abs(g5[b].scores[0] - sf) / max(1.0, abs(sf)))
        lang, al = [], []
        if cfg.is_multilingual:
            gl = model.detect_language(enc)
            rl = oracle.detect_language(sub)
            for j, b in enumerate(SUB):
                gp = dict(gl[b])
                lang.append(max(abs(gp[names[tid - cfg.lang_begin]] - p) for tid, p in rl[j]))
        text = [[t for t in g.sequences_ids[0] if t < cfg.eot] for g in g1]
        ga = model.align(enc, cfg.sot_sequence, text, [3000] * B, median_filter_width=7)
        ra = oracle.align(sub, cfg.sot_sequence, [text[b] for b in SUB], [3000] * 2, median_filter_width=7)
        al = [float(np.abs(np.array(ga[b].text_token_probs) - np.array(ra[j].text_token_probs)).max())
              for j, b in enumerate(SUB)]
        kwt = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 100, suppress_tokens=sup,
                   min_new_tokens=100)
        model.generate(enc, [prompt] * B, **kwt)
        t0 = time.perf_counter()
        for _ in range(3):
            model.generate(enc, [prompt] * B, **kwt)
        dt = (time.perf_counter() - t0) / 3
        model.generate(model.encode_pcm(chunks[:1]), [prompt], **kwt)
        t0 = time.perf_counter()
        for _ in range(3):
            model.generate(model.encode_pcm(chunks[:1]), [prompt], **kwt)
        d1 = (time.perf_counter() - t0) / 3
        f = lambda v: " / ".join(f"{x:.2e}" for x in v)   # noqa: E731
        print(f"[{name}] FWAMD_LN_UNFOLD={order}: per-token tf {f(tf)}; beam-48 score {f(bs)}; language prob {f(lang)}; "
              f"align prob {f(al)}; solo 16-chunk x 100-step decode {dt * 1e3:.0f} ms; single utterance encode+decode "
              f"{d1 * 1e3:.0f} ms", flush=True)
        model.unload_model()
        del model, enc


if __name__ == "__main__":
    main()
