"""Probe: do TWO concurrent decode runs (two independent decode groups on one GPU, each with half the workers) beat ONE
merged run with all the workers?  The cross-attention stream is HBM-bound, the decoder linears are not: two runs on two
streams could overlap them.  Same workload as bench.py (large-v3 fp16, 16 chunks x beam 5, 100 tokens), PCM resident.
    python profiles/two_groups_probe.py [steps_per_group]"""
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faster_whisper_amd import Whisper, get_config, synthetic_weights  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = get_config("large-v3")
    w = synthetic_weights(cfg, seed=1234)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 100, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup, min_new_tokens=100)
    chunks = bench.synth_chunks(16, seed=1000)

    def run(groups, workers_per_group, n_steps):
        models = [Whisper("synthetic:large-v3", device="cuda", files={"config": cfg, "weights": w}, max_batch_size=16,
                          max_beam_size=5, inter_threads=workers_per_group) for _ in range(groups)]
        staged = [m.stage_pcm(chunks) for m in models]
        pools = [ThreadPoolExecutor(max_workers=workers_per_group) for _ in models]

        def step(g):
            return models[g].generate(models[g].encode_pcm_staged(staged[g]), [prompt] * 16, **kw)

        for g in range(groups):                                  # warm: every worker once
            list(pools[g].map(lambda _: step(g), range(workers_per_group)))
        for m in models:
            m.synchronize()
        t0 = time.perf_counter()
        futs = [pools[g].submit(step, g) for _ in range(n_steps) for g in range(groups)]
        for f in futs:
            f.result()
        for m in models:
            m.synchronize()
        dt = time.perf_counter() - t0
        total = n_steps * groups
        stats = [m.decode_stats() for m in models]
        print(f"{groups} group(s) x {workers_per_group} workers, {total} steps: {30.0 * 16 * total / dt:.1f}x, "
              f"{1e3 * dt / total:.1f} ms per step; runs {[s['runs'] for s in stats]}, largest {[s['max_run_chunks'] for s in stats]}",
              flush=True)
        for m, s in zip(models, staged):
            m.free_staged(s)
            m.unload_model()
        for p in pools:
            p.shutdown()

    run(1, 32, 2 * steps)
    run(2, 16, steps)
    run(2, 24, steps)
    run(1, 32, 2 * steps)


if __name__ == "__main__":
    main()
