"""Round 5 A/B harness: ONE process, ONE box, one loaded model — boxes differ by 3-5 %, more than most kernel changes.

Interleaves the settings of a measurement knob (fw_test_knob id, values) and measures, for each setting and round:
  * the single-utterance latency (config C2, best of 3),
  * one 16-chunk batch at a time (solo decode runs, mean of 3),
  * the merged steady state: `--steps` batches over `--workers` host threads (two decode lanes), the bench's own loop,
  * optionally one profiled round (HIP events per kernel family, one lane).

    python profiles/ab_r05.py --knob 2 --values 1,0 --rounds 2 [--profile] [--steps 64]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", type=int, default=2)
    ap.add_argument("--values", default="1,0")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--no-merged", action="store_true")
    ap.add_argument("--env", default=None, help="NAME: the settings are values of this environment variable, read by the "
                                               "engine at every call (instead of a knob)")
    a = ap.parse_args()
    from concurrent.futures import ThreadPoolExecutor
    from faster_whisper_amd import _lib, get_config
    args = bench.parse_args(["--workers", str(a.workers), "--steps", str(a.steps)])
    cfg = get_config(args.model)
    model, _ = bench.build_backend(args, cfg, 0, 1, 0)
    lib = _lib.load()
    chunks = bench.synth_chunks(args.batch, seed=1000)
    staged = model.stage_pcm(chunks)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    L = args.new_tokens
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=args.beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup, min_new_tokens=L)
    pool = ThreadPoolExecutor(max_workers=a.workers)

    def step():
        return model.generate(model.encode_pcm_staged(staged), [prompt] * args.batch, **kw)

    def merged(n):
        model.synchronize()
        t0 = time.perf_counter()
        outs = [f.result() for f in [pool.submit(step) for _ in range(n)]]
        model.synchronize()
        return time.perf_counter() - t0, outs[-1]

    def setting(v):
        if a.env:
            os.environ[a.env] = str(v)
        else:
            _lib.check(lib.fw_test_knob(a.knob, int(v)))

    values = a.values.split(",")
    ref = None
    merged(a.workers)     # warm every worker
    for rnd in range(a.rounds):
        for v in values:
            setting(v)
            rec = {"round": rnd, "setting": v}
            rec["single_utterance_ms"] = bench.single_utterance(model, cfg, chunks[0], prompt, kw, L).get("latency_ms")
            ob = bench.one_batch(model, staged, chunks, prompt, kw, L, args.batch, reps=3)
            rec["one_batch_ms"] = ob.get("latency_ms_per_batch")
            rec["c4_rank_batch_ms"] = (ob.get("c4_rank_batch") or {}).get("latency_ms")
            if not a.no_merged:
                merged(a.workers)
                dt, last = merged(a.steps)
                rec["merged_rtf"] = round(30.0 * args.batch * a.steps / dt, 1)
                rec["merged_ms_per_step"] = round(1e3 * dt / a.steps, 2)
                sig = [(r.sequences_ids, r.scores, r.no_speech_prob) for r in last]
                if ref is None:
                    ref = sig
                rec["same_results_as_first_setting"] = sig == ref
            if a.profile:
                model.set_decode_lanes(1)
                model.profile(True, replica=None)
                [f.result() for f in [pool.submit(step) for _ in range(a.workers)]]
                model.synchronize()
                rep = model.profile_report(replica=None)
                model.profile(False, replica=None)
                model.set_decode_lanes(2)
                rec["families_ms_per_batch"] = {k: round(x["ms"] / a.workers, 3) for k, x in rep.items()}
                # ... and of ONE batch decoded alone (solo run)
                model.profile(True, replica=None)
                step()
                model.synchronize()
                rep = model.profile_report(replica=None)
                model.profile(False, replica=None)
                rec["families_ms_solo_batch"] = {k: round(x["ms"], 3) for k, x in rep.items()}
            print(json.dumps(rec), flush=True)
    setting(values[0] if a.env else 0)
    model.free_staged(staged)
    pool.shutdown()


if __name__ == "__main__":
    main()
