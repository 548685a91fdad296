"""Where the VAD front of config C5 spends its time for a long recording (the part of `pipeline` that runs BEFORE the first
chunk is transcribed): numpy framing, the device network (front-end kernel + sequential LSTM kernel + copies), the Python
state machine of get_speech_timestamps, collect_chunks.

    python profiles/vad_bench.py [hours]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402


def main():
    hours = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    from faster_whisper_amd import vad as fvad
    n_chunks = int(hours * 120)
    chunks = bench.synth_chunks(8, seed=2000)
    audio = np.concatenate([chunks[i % 8] for i in range(n_chunks)])
    for i in range(n_chunks):
        audio[480000 * i + 440000:480000 * (i + 1)] = 0.0
    model = bench.synthetic_vad(0)
    model(np.zeros(512 * 64, np.float32))          # warm
    out = {"hours": hours, "windows": len(audio) // 512 + 1}
    t0 = time.perf_counter()
    padded = np.pad(audio, (0, 512 - len(audio) % 512))
    out["np_pad_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    probs = model(padded)
    out["model_call_s"] = round(time.perf_counter() - t0, 3)
    if hasattr(model, "last_timing"):
        out["model_call_parts"] = model.last_timing
    opts = fvad.VadOptions(max_speech_duration_s=30, min_silence_duration_ms=160)
    t0 = time.perf_counter()
    spans = fvad.get_speech_timestamps(audio, opts, speech_probs=probs)
    out["state_machine_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    ch, meta = fvad.collect_chunks(audio, spans, max_duration=30)
    out["collect_chunks_s"] = round(time.perf_counter() - t0, 3)
    out["spans"], out["chunks"] = len(spans), len(ch)
    fvad._VAD_MODEL = model
    t0 = time.perf_counter()
    spans2 = fvad.get_speech_timestamps(audio, opts)
    out["get_speech_timestamps_total_s"] = round(time.perf_counter() - t0, 3)
    assert spans2 == spans
    print(json.dumps(out))


if __name__ == "__main__":
    main()
