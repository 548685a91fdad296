"""TEST INFRASTRUCTURE — the inputs of the host-logic parity fixtures (tests/golden/host_*.json).

Shared by oracle/gen_golden_host.py (runs the REFERENCE's host code on these inputs, build container only) and
tests/test_host_golden.py (runs this repository's host code on the same inputs).  Audio is synthetic and
seeded; the "model" is oracle/scripted_backend.py.
"""
import numpy as np

SR = 16000


def synth_audio(seed: int, seconds: float, silences=()) -> np.ndarray:
    """band noise + three partials (SURVEY.md section 8d recipe) with exact-zero gaps [(start_s, end_s), ...]"""
    rng = np.random.default_rng(seed)
    n = int(seconds * SR)
    t = np.arange(n) / SR
    x = 0.1 * rng.standard_normal(n)
    for f in (220.0, 440.0, 880.0):
        x += 0.05 * np.sin(2 * np.pi * f * t)
    # slow loudness drift so that different windows have different fingerprints
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * t / 17.0) ** 2
    for a, b in silences:
        x[int(a * SR):int(b * SR)] = 0.0
    return x.astype(np.float32)


def speech_probs(padded_audio: np.ndarray) -> np.ndarray:
    """scripted stand-in for the Silero network: one probability per 512-sample window from the window's RMS
    (loud -> 0.92, silent -> 0.03, in between -> linear), same contract as SileroVADModel.__call__"""
    w = np.asarray(padded_audio, dtype=np.float64).reshape(-1, 512)
    rms = np.sqrt((w ** 2).mean(axis=1))
    return np.clip(0.03 + (rms / 0.05) * 0.89, 0.03, 0.92).astype(np.float32)


# name -> dict(kind, audio=(seed, seconds, silences), kwargs=transcribe arguments)
SCENARIOS = {
    "seq_default": dict(kind="sequential", audio=(1, 75.0, ((20.0, 27.5), (50.0, 58.0))),
                        kwargs=dict(language="en")),
    "seq_words_hallucination": dict(kind="sequential", audio=(2, 64.0, ((8.0, 14.0), (40.0, 47.0))),
                                    kwargs=dict(language="en", word_timestamps=True,
                                                hallucination_silence_threshold=1.0)),
    "seq_words_zh": dict(kind="sequential", audio=(3, 41.0, ()),
                         kwargs=dict(language="zh", word_timestamps=True, temperature=0.0)),
    "seq_clips_prompt": dict(kind="sequential", audio=(4, 70.0, ((31.0, 33.0),)),
                             kwargs=dict(language="de", without_timestamps=True, clip_timestamps="5,20,30,50",
                                         initial_prompt="hello world", prefix="the model", hotwords="whisper audio",
                                         max_new_tokens=60, condition_on_previous_text=False)),
    "seq_clip_list_odd": dict(kind="sequential", audio=(5, 50.0, ()),
                              kwargs=dict(language="en", clip_timestamps=[12.0, 25.5, 31.0],
                                          temperature=[0.0, 0.4, 0.8], best_of=3, beam_size=2,
                                          compression_ratio_threshold=2.0, log_prob_threshold=-0.9,
                                          prompt_reset_on_temperature=0.3, initial_prompt=[260, 261, 262])),
    "seq_detect_multilingual": dict(kind="sequential", audio=(6, 45.0, ()),
                                    kwargs=dict(language=None, multilingual=True, task="translate",
                                                language_detection_segments=2, language_detection_threshold=0.7,
                                                no_speech_threshold=None)),
    "seq_vad": dict(kind="sequential", audio=(7, 80.0, ((10.0, 18.0), (30.0, 31.0), (55.0, 70.0))),
                    kwargs=dict(language="en", vad_filter=True, word_timestamps=True,
                                vad_parameters=dict(min_silence_duration_ms=500, speech_pad_ms=200))),
    "seq_words_hal_b": dict(kind="sequential", audio=(11, 90.0, ((3.0, 9.0), (33.0, 36.0), (61.0, 75.0))),
                            kwargs=dict(language="en", word_timestamps=True, hallucination_silence_threshold=0.5,
                                        temperature=[0.0, 0.2, 0.4])),
    "seq_words_hal_c": dict(kind="sequential", audio=(12, 58.0, ((0.0, 4.0), (25.0, 31.0))),
                            kwargs=dict(language="es", word_timestamps=True, hallucination_silence_threshold=2.0,
                                        condition_on_previous_text=False, log_prob_threshold=None)),
    "seq_all_fail": dict(kind="sequential", audio=(13, 62.0, ()),
                         kwargs=dict(language="en", temperature=[0.0, 0.2, 0.6], compression_ratio_threshold=1.05,
                                     log_prob_threshold=-0.05, no_speech_threshold=0.9)),
    "seq_silence": dict(kind="sequential", audio=(14, 95.0, ((0.0, 31.0), (60.0, 95.0))),
                        kwargs=dict(language="en", no_speech_threshold=0.5, log_prob_threshold=-1.0,
                                    compression_ratio_threshold=None)),
    "bat_clips_words": dict(kind="batched", audio=(8, 100.0, ((44.0, 47.0),)),
                            kwargs=dict(language="en", word_timestamps=True, batch_size=3, without_timestamps=False,
                                        clip_timestamps=[dict(start=0.0, end=28.0), dict(start=28.0, end=44.0),
                                                         dict(start=47.0, end=77.0), dict(start=77.0, end=100.0)])),
    "bat_vad": dict(kind="batched", audio=(9, 130.0, ((12.0, 15.0), (40.0, 52.0), (90.0, 91.0), (118.0, 130.0))),
                    kwargs=dict(language=None, vad_filter=True, batch_size=2, word_timestamps=True,
                                vad_parameters=dict(min_silence_duration_ms=400))),
    "bat_short_multilingual": dict(kind="batched", audio=(10, 21.0, ()),
                                   kwargs=dict(language=None, multilingual=True, vad_filter=False, batch_size=4,
                                               initial_prompt="the test", hotwords="speech model",
                                               max_new_tokens=40)),
}

# inputs of the unit-level fixtures
VAD_CASES = {
    "default": dict(),
    "tight": dict(threshold=0.6, min_speech_duration_ms=250, min_silence_duration_ms=100, speech_pad_ms=30),
    "max30": dict(max_speech_duration_s=8.0, min_silence_duration_ms=160, speech_pad_ms=400),
    "neg": dict(threshold=0.5, neg_threshold=0.1, min_silence_duration_ms=300, speech_pad_ms=0,
                max_speech_duration_s=5.0),
}


def vad_prob_tracks():
    """name -> (n_audio_samples, probabilities) hand-shaped + random tracks for the state machine"""
    rng = np.random.default_rng(99)
    tracks = {}
    p = np.full(600, 0.02)
    p[20:200] = 0.9
    p[203:206] = 0.3                  # short dip inside speech (between neg and threshold)
    p[260:300] = 0.8
    p[300:304] = 0.05
    p[304:420] = 0.95
    p[480:490] = 0.7                  # short burst
    tracks["hand"] = (600 * 512 - 100, p)
    tracks["all_speech"] = (400 * 512, np.full(400, 0.9))
    tracks["silence"] = (100 * 512 - 7, np.full(100, 0.01))
    tracks["open_end"] = (300 * 512 - 256, np.r_[np.full(100, 0.02), np.full(200, 0.88)])
    walk = np.clip(0.5 + np.cumsum(rng.normal(0, 0.12, 2000)) * 0.3, 0, 1)
    tracks["random_walk"] = (2000 * 512 - 33, walk)
    tracks["noisy"] = (1500 * 512, np.clip(rng.random(1500) * (np.sin(np.arange(1500) / 40.0) > 0) + 0.02, 0, 1))
    return tracks


SPLIT_TEXTS = {
    "en": [" hello world, this is a test.", "hello (the model) was \"not\" ready!", " 世界 café and - time",
           " I: they [ whisper ]", ""],
    "zh": [" 世界 hello 世界", "café 世", " the test."],
}
