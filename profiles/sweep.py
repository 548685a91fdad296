"""Experiment driver for engine knobs (FWAMD_* environment variables are read once per process, so one
process = one configuration).  Times the bench.py workload single-stream and with 8 batches in flight in the
same process, reusing a packed weight blob cached in /tmp so that repeated configurations do not pay the
synthetic-weight generation again.

    FWAMD_GEMM_DEPTH=4 python profiles/sweep.py --tag depth4 >> gpurun_out/sweep.jsonl

Not part of the product path; bench.py stays the only source of the reported metric."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--new-tokens", type=int, default=100)
    ap.add_argument("--workers", default="1,8")
    ap.add_argument("--steps-per-worker", type=int, default=3)
    ap.add_argument("--compute-type", default="float16")
    args = ap.parse_args()

    import torch
    from bench import synth_chunks
    from faster_whisper_amd import Whisper, get_config, pack_blob, synthetic_weights

    cfg = get_config(args.model)
    # the decoder weight layout is decided at pack time (FWAMD_DEC_GEMM), so it is part of the cache key
    cache = (f"/tmp/fwamd_blob_{args.model}_{args.compute_type}_{os.environ.get('FWAMD_DEC_GEMM', 'default')}"
             f"_{os.environ.get('FWAMD_DEC_GEMM_I8', 'default')}.npy")
    t0 = time.time()
    if os.path.exists(cache):
        blob = np.load(cache, mmap_mode="r")
    else:
        blob = pack_blob(cfg, synthetic_weights(cfg, seed=1234), 1 if args.compute_type == "int8_float16" else 0)
        np.save(cache, blob)
    dev_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
    wmax = max(int(w) for w in args.workers.split(","))
    model = Whisper(f"synthetic:{args.model}", device="cuda", max_batch_size=args.batch, max_beam_size=args.beam,
                    inter_threads=wmax, compute_type=args.compute_type,
                    blob_dev=(dev_blob.data_ptr(), dev_blob.numel()))
    load_s = time.time() - t0
    staged = model.stage_pcm(synth_chunks(args.batch, seed=1000))
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    L = args.new_tokens
    kw = dict(beam_size=args.beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, min_new_tokens=L,
              suppress_tokens=[cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe])

    def step(_=None):
        return model.generate(model.encode_pcm_staged(staged), [prompt] * args.batch, **kw)

    def sync():
        for r in model._replicas:
            model._lib.fw_synchronize(r.handle)

    out = {"tag": args.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("FWAMD_")},
           "load_s": round(load_s, 1), "rtf": {}}
    for w in [int(x) for x in args.workers.split(",")]:
        pool = ThreadPoolExecutor(max_workers=w)
        list(pool.map(step, range(w)))          # warm-up: one step per worker thread (graph capture etc.)
        sync()
        n = args.steps_per_worker * w
        t1 = time.perf_counter()
        res = list(pool.map(step, range(n)))
        sync()
        dt = time.perf_counter() - t1
        assert all(len(r.sequences_ids[0]) == L for r in res[-1])
        out["rtf"][str(w)] = round(30.0 * args.batch * n / dt, 1)
        pool.shutdown()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
