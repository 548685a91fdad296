"""Diagnostic for tests/test_gpu_int8.py::test_generate_int8[tiny.en-*]: the engine prefers token B where the oracle
prefers token A at step 1 by more than the margin.  Which side's log-prob of which token is off?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import bench_audio, forced_score, make_model  # noqa: E402
from faster_whisper_amd.backend import StorageView  # noqa: E402
from oracle.whisper import OracleWhisper  # noqa: E402

cfg, w, model = make_model("tiny.en", seed=11, max_batch=4, max_beam=5, compute_type="int8_float16")
oracle = OracleWhisper(cfg, w, int8=True)
chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
feats = model.log_mel(chunks)
enc = model.encode(StorageView.from_array(feats))
enc_np = enc.to_numpy()
prompt = list(cfg.sot_sequence)
sup = sorted({cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe, 1, 2, 7})
for L in (2, 3):
    kw = dict(beam_size=1, max_length=len(prompt) + L, suppress_blank=True, suppress_tokens=sup, max_initial_timestamp_index=50,
              length_penalty=0.0)
    got = model.generate(enc, [prompt] * 3, return_scores=True, **kw)
    ref = oracle.generate(enc_np, [prompt] * 3, **kw)
    for b in range(3):
        ids_e, ids_o = got[b].sequences_ids[0], ref[b].sequences_ids[0]
        print(f"L={L} chunk {b}: engine {ids_e} score {got[b].scores[0]:.5f} | oracle {ids_o} score {ref[b].scores[0]:.5f} margins {np.round(ref[b].margins, 4)}")
        print(f"      oracle's score of the engine's ids {forced_score(oracle, enc_np[b], prompt, ids_e, kw):.5f}")
        if ids_e != ids_o:
            # make the engine take the oracle's token at the first differing step by suppressing its own choice
            k = next(i for i in range(L) if ids_e[i] != ids_o[i])
            kw2 = dict(kw, suppress_tokens=sup + [ids_e[k]])
            g2 = model.generate(enc, [prompt] * 3, return_scores=True, **kw2)[b]
            print(f"      engine with {ids_e[k]} suppressed: {g2.sequences_ids[0]} score {g2.scores[0]:.5f}; "
                  f"oracle's score of those ids (same suppression) {forced_score(oracle, enc_np[b], prompt, g2.sequences_ids[0], kw2):.5f}")
            kw3 = dict(kw, suppress_tokens=sup + [ids_o[k]])
            r3 = oracle.generate(enc_np[b:b + 1], [prompt], **kw3)[0]
            print(f"      oracle with {ids_o[k]} suppressed: {r3.sequences_ids[0]} score {r3.scores[0]:.5f}")
