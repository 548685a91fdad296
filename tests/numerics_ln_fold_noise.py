"""How far apart are two valid fp16 evaluation orders of large-v3 (synthetic weights)?  CPU only, ~8 minutes.

    python tests/numerics_ln_fold_noise.py > tests/golden/numerics_ln_fold_noise.txt

Orders compared on the same encoder output, same token ids (teacher forcing):
    o32   float32 throughout                                   (what the reference's CPU path computes)
    o16   fp16 storage, explicit decoder LayerNorms            (what the reference's fp16 path computes)
    fold  fp16 storage, decoder LayerNorms folded into the consuming linears, W' = fp16(W * g)
          (what the engine computes: engine.hip add_folded, dec_kernels.hip dec_gemm_frag_kernel<LNF>)
The output is the justification of the tolerances in tests/test_gpu_full_size.py: the engine is compared tightly
with `fold`, and with `o16` only at the level at which `fold` and `o16` (and `o16` and `o32`) differ here.
Not a test (pytest does not collect it): it measures, it asserts nothing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import bench_audio  # noqa: E402
from faster_whisper_amd import get_config, synthetic_weights  # noqa: E402
from oracle import logmel as olm  # noqa: E402
from oracle.whisper import OracleWhisper  # noqa: E402


def main(n=6):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = get_config("large-v3")
    w = synthetic_weights(cfg, seed=1234)
    o16 = OracleWhisper(cfg, w, emulate_fp16=True)
    o32 = OracleWhisper(cfg, w, emulate_fp16=False)
    t0 = time.time()
    feats = olm.log_mel_chunks([bench_audio(480000, seed=100 + 3 * i) for i in range(n)], cfg.n_mels)
    enc = np.concatenate([o16.encode(feats[i:i + 2]) for i in range(0, n, 2)])
    print(f"# {n} chunks encoded in {time.time() - t0:.0f} s", flush=True)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=1, max_length=len(prompt) + 8, length_penalty=0.0, suppress_tokens=sup)
    r16 = o16.generate(enc, [prompt] * n, **kw)
    ids = [r.sequences_ids[0] for r in r16]
    r32 = o32.generate(enc, [prompt] * n, force_tokens=ids, **kw)
    l16 = [dict(r) for r in o16.detect_language(enc)]
    l32 = [dict(r) for r in o32.detect_language(enc)]
    kw5 = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 6, suppress_tokens=sup)
    b16 = o16.generate(enc[:3], [prompt] * 3, **kw5)
    o16.fold_ln = True
    rf = o16.generate(enc, [prompt] * n, force_tokens=ids, **kw)
    lf = [dict(r) for r in o16.detect_language(enc)]
    bf = o16.generate(enc[:3], [prompt] * 3, **kw5)
    print("# 8 teacher-forced steps, |difference of the cumulative log-prob| / 8 tokens")
    for j in range(n):
        a, b, c = r16[j].scores[0], rf[j].scores[0], r32[j].scores[0]
        print(f"chunk {j}: o16-o32 {abs(a - c) / 8:.2e}   fold-o32 {abs(b - c) / 8:.2e}   fold-o16 {abs(a - b) / 8:.2e}")
    print("# detect_language, max |difference of a language probability|")
    for j in range(n):
        d1 = max(abs(l16[j][k] - l32[j][k]) for k in l16[j])
        d2 = max(abs(lf[j][k] - l32[j][k]) for k in l16[j])
        d3 = max(abs(lf[j][k] - l16[j][k]) for k in l16[j])
        print(f"chunk {j}: o16-o32 {d1:.2e}   fold-o32 {d2:.2e}   fold-o16 {d3:.2e}")
    print("# beam 5, 6 steps: same best hypothesis?  |difference of its score|")
    for j in range(3):
        print(f"chunk {j}: fold-o16 same ids {b16[j].sequences_ids[0] == bf[j].sequences_ids[0]}   "
              f"{abs(b16[j].scores[0] - bf[j].scores[0]):.2e}")


if __name__ == "__main__":
    main()
