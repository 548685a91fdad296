"""Round 6 (second session), verdict item 4: would MORE than two decode lanes help the lone batch (C4's per-rank batch)?

A lone 15-chunk batch is already split into two half batches on the two lanes of its decode group (316 -> 297 ms).  Before
generalising the group to N lanes, the question is priced with what exists: TWO models built over ONE weight blob in HBM
(each its own decode group of two lanes, own streams, the same weights) give four concurrent decode runs.

    one call | two halves (model A) | three thirds (A, A, B) | four quarters (A, A, B, B)      -> ms, best of 3 after a warm-up

    python profiles/ab_r06_quarters.py >> profiles/r06_ab_quarters.jsonl
"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402


def main():
    import torch
    from faster_whisper_amd import Whisper, get_config, pack_blob, synthetic_weights
    args = bench.parse_args([])
    cfg = get_config(args.model)
    blob = pack_blob(cfg, synthetic_weights(cfg, seed=1234), 0)
    dev_blob = torch.from_numpy(blob).cuda()
    del blob
    chunks = bench.synth_chunks(args.batch, seed=1000)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    L = args.new_tokens
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=args.beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup, min_new_tokens=L)
    models = [Whisper(f"synthetic:{args.model}", device="cuda", blob_dev=(dev_blob.data_ptr(), dev_blob.numel()),
                      max_batch_size=args.batch, max_beam_size=args.beam, inter_threads=4) for _ in range(2)]

    def run(parts, owners):
        out = [None] * len(parts)

        def one(i):
            m = models[owners[i]]
            out[i] = m.generate(m.encode_pcm(parts[i]), [prompt] * len(parts[i]), **kw)

        best = None
        for rep in range(4):                 # (first: warm — graphs of the shapes)
            ths = [threading.Thread(target=one, args=(i,)) for i in range(len(parts))]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            d = time.perf_counter() - t0
            if rep:
                best = d if best is None else min(best, d)
        ids = [r.sequences_ids for part in out for r in part]
        return best, ids

    def split(c, n):
        q, r = divmod(len(c), n)
        parts, i = [], 0
        for k in range(n):
            sz = q + (1 if k < r else 0)
            parts.append(c[i:i + sz])
            i += sz
        return parts

    for n_chunks in (15, 16, 8):
        c = chunks[:n_chunks]
        rec = {"chunks": n_chunks}
        ref = None
        for name, n, owners in (("one_call", 1, [0]), ("two_halves_A", 2, [0, 0]), ("two_halves_A_B", 2, [0, 1]),
                                ("three_thirds_AAB", 3, [0, 0, 1]), ("four_quarters_AABB", 4, [0, 0, 1, 1])):
            dt, ids = run(split(c, n), owners)
            if ref is None:
                ref = ids
            rec[name + "_ms"] = round(1e3 * dt, 1)
            rec[name + "_same_ids"] = ids == ref
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
