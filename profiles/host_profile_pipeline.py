"""Where does the host spend the time between the engine and the last Segment?  (round 6, verdict item 2: distil-large-v3's
`pipeline` is 26 % below its hot path.)

Runs bench.pipeline_rtf's call (BatchedInferencePipeline.transcribe on a synthetic recording) with a SAMPLING profiler over
every Python thread: a background thread reads sys._current_frames() every few milliseconds and counts, per thread role
(the consumer = the thread that iterates the segments; workers = the pool threads that run decode_batch), the innermost
frames and the frames of this repository on the stack (inclusive).  Blocked-in-C samples (ctypes call into libfwamd, lock
waits) show up under the Python function that made the call.

    python profiles/host_profile_pipeline.py --model distil-large-v3 --chunks 480 --word-timestamps [--vad]
"""
import argparse
import collections
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.period = period
        self.stop = False
        self.inner = collections.Counter()      # (role, file:func) of the innermost frame
        self.incl = collections.Counter()       # (role, file:func) anywhere on the stack (once per sample)
        self.n = collections.Counter()          # samples per role
        self.consumer = threading.get_ident()

    def run(self):
        me = threading.get_ident()
        while not self.stop:
            for tid, fr in sys._current_frames().items():
                if tid == me:
                    continue
                role = "consumer" if tid == self.consumer else "worker"
                seen = set()
                f = fr
                first = True
                while f is not None:
                    co = f.f_code
                    key = f"{os.path.basename(co.co_filename)}:{co.co_name}"
                    if first:
                        self.inner[(role, key)] += 1
                        first = False
                    if key not in seen:
                        seen.add(key)
                        self.incl[(role, key)] += 1
                    f = f.f_back
                self.n[role] += 1
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="distil-large-v3")
    ap.add_argument("--chunks", type=int, default=480)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--word-timestamps", action="store_true")
    ap.add_argument("--vad", action="store_true")
    a = ap.parse_args()
    from faster_whisper_amd import get_config
    args = bench.parse_args(["--model", a.model, "--workers", str(a.workers)])
    cfg = get_config(a.model)
    model, _ = bench.build_backend(args, cfg, 0, 1, 0)
    # un-profiled reference run, then the sampled one
    base = bench.pipeline_rtf(model, cfg, a.chunks, args.batch, args.beam, args.new_tokens,
                              word_timestamps=a.word_timestamps, vad=a.vad)
    s = Sampler()
    s.start()
    prof = bench.pipeline_rtf(model, cfg, a.chunks, args.batch, args.beam, args.new_tokens,
                              word_timestamps=a.word_timestamps, vad=a.vad)
    s.stop = True
    s.join()
    out = {"model": a.model, "chunks": a.chunks, "word_timestamps": a.word_timestamps, "vad": a.vad,
           "pipeline_unprofiled": base, "pipeline_sampled": prof, "samples": dict(s.n)}
    for role in ("consumer", "worker"):
        tot = max(1, s.n[role])
        out[f"{role}_innermost_top"] = [(k[1], round(v / tot, 3)) for k, v in s.inner.most_common(60) if k[0] == role][:14]
        out[f"{role}_inclusive_top"] = [(k[1], round(v / tot, 3)) for k, v in s.incl.most_common(120) if k[0] == role][:24]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
