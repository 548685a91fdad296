"""Round 6 (second session): do THREE or FOUR decode lanes overlap the two roofs better than two?

Two lanes (two decode runs in flight, each a serial chain of kernels) leave the chip with at most two decode kernels and
one encoder kernel in flight: the cross-attention stream averages 2.75 TB/s over a step (65 ms at 6.1 TB/s in 144 ms of
wall).  ONE process, ONE box, one model built with FWAMD_DECODE_LANES=4; the lanes allowed in flight are switched with
fw_model_set_decode_lanes between measurements (same kernels, same weights, same results: ids compared):

    for lanes in 2, 3, 4, 2, 3, 4:  steady state (--steps batches over --workers workers) and the driver's 20-step burst

    FWAMD_DECODE_LANES=4 python profiles/ab_r06_lanes.py [--workers 32,48] >> profiles/r06_ab_lanes.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FWAMD_DECODE_LANES", "4")

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--workers", default="32")
    ap.add_argument("--lanes", default="2,3,4,2,3,4")
    a = ap.parse_args()
    import torch
    from concurrent.futures import ThreadPoolExecutor
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
    ref_ids = None
    for W in [int(x) for x in a.workers.split(",")]:
        model = Whisper(f"synthetic:{args.model}", device="cuda", blob_dev=(dev_blob.data_ptr(), dev_blob.numel()),
                        max_batch_size=args.batch, max_beam_size=args.beam, inter_threads=W)
        staged = model.stage_pcm(chunks)
        pool = ThreadPoolExecutor(max_workers=W)

        def step():
            return model.generate(model.encode_pcm_staged(staged), [prompt] * args.batch, **kw)

        def merged(n):
            model.synchronize()
            t0 = time.perf_counter()
            outs = [f.result() for f in [pool.submit(step) for _ in range(n)]]
            model.synchronize()
            return time.perf_counter() - t0, outs[-1]

        merged(W)                         # warm: graphs, pools
        for lanes in [int(x) for x in a.lanes.split(",")]:
            model.set_decode_lanes(lanes)
            merged(W)                     # settle into the setting
            s0 = model.decode_stats() if hasattr(model, "decode_stats") else None
            dt, res = merged(a.steps)
            s1 = model.decode_stats() if hasattr(model, "decode_stats") else None
            ids = [r.sequences_ids for r in res]
            if ref_ids is None:
                ref_ids = ids
            bt, _ = merged(20)            # the driver's command: one burst of 20 batches
            rec = {"workers": W, "lanes": lanes, "steady_x": round(30.0 * args.batch * a.steps / dt, 1),
                   "ms_per_step": round(1e3 * dt / a.steps, 2), "burst20_x": round(30.0 * args.batch * 20 / bt, 1),
                   "same_ids_as_first_setting": ids == ref_ids}
            if s0 is not None and s1 is not None:
                try:
                    rec["decode_runs"] = int(s1["runs"] - s0["runs"])
                    rec["chunks_per_run"] = round((s1["chunks"] - s0["chunks"]) / max(1, s1["runs"] - s0["runs"]), 1)
                    rec["pool_chunks"] = s1["decode_batch"]
                    rec["run_capacity"] = s1["run_capacity"]
                except Exception:
                    pass
            print(json.dumps(rec), flush=True)
        pool.shutdown()
        model.free_staged(staged)
        model.unload_model()
        del model


if __name__ == "__main__":
    main()
