"""Parity at BASELINE.json's full size (large-v3 geometry, synthetic weights): the engine against the fp16-emulating
oracle on one 30 s chunk — encoder output, a few teacher-forced greedy steps, language probabilities.  Takes about
a minute (22 s of weight generation + ~20 s of CPU oracle on the box's cores).

Written after round 1's GPU budget was spent, so the tolerances below are estimates from the small geometries:
opt-in (FWAMD_TEST_UNVALIDATED=1) until it has been seen passing on hardware."""
import os

import numpy as np
import pytest

from conftest import bench_audio

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("FWAMD_TEST_UNVALIDATED") != "1",
                                 reason="full-size parity test not yet calibrated on hardware")]


def test_large_v3_against_oracle():
    import torch
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    from faster_whisper_amd.backend import StorageView
    from oracle.whisper import OracleWhisper
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_config("large-v3")
    w = synthetic_weights(cfg, seed=1234)
    model = Whisper("synthetic:large-v3", device="cuda", files={"config": cfg, "weights": w}, max_batch_size=1,
                    max_beam_size=5)
    oracle = OracleWhisper(cfg, w, emulate_fp16=True)
    feats = model.log_mel([bench_audio(480000, seed=5)])
    enc = model.encode(StorageView.from_array(feats))
    got = enc.to_numpy()
    ref = oracle.encode(feats)
    rel = float(np.abs(got - ref).max() / np.abs(ref).max())
    rms = float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    print(f"large-v3 encoder: max rel {rel:.2e}, rms rel {rms:.2e}")
    assert rel < 3e-2 and rms < 5e-3
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=1, max_length=len(prompt) + 6, length_penalty=0.0,
              suppress_tokens=[cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe])
    g = model.generate(enc, [prompt], return_scores=True, return_no_speech_prob=True, **kw)[0]
    r = oracle.generate(got, [prompt], force_tokens=[g.sequences_ids[0]], **kw)[0]   # oracle fed the engine's encoder output
    print(f"large-v3 teacher-forced cum logprob {g.scores[0]:.5f} vs {r.scores[0]:.5f}; "
          f"no_speech {g.no_speech_prob:.3e} vs {r.no_speech_prob:.3e}")
    assert r.sequences_ids[0] == g.sequences_ids[0]
    assert abs(g.scores[0] - r.scores[0]) < 5e-3 * max(1.0, abs(r.scores[0]))
    assert abs(g.no_speech_prob - r.no_speech_prob) < 2e-3
    gl = dict(model.detect_language(enc)[0])
    names = None
    from faster_whisper_amd.backend import language_token_strings
    names = language_token_strings(cfg)
    for tid, p in oracle.detect_language(got)[0][:5]:
        assert abs(gl[names[tid - cfg.lang_begin]] - p) < 2e-3
