"""TEST INFRASTRUCTURE — differential fuzzing of the host logic against the REFERENCE's own code (build container
only: needs /root/reference).  For each seed a random recording (random silences) and a random combination of
transcribe() arguments is run through the reference's WhisperModel / BatchedInferencePipeline and through this
repository's, both on oracle/scripted_backend.py; segments, words, info and the backend call logs must be equal.

    python oracle/fuzz_host.py --seeds 200          # prints the first mismatch, exit code 1 on any
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_case(seed):
    rng = np.random.default_rng(seed)
    seconds = float(rng.uniform(8, 95))
    silences, t = [], 0.0
    while True:
        t += float(rng.uniform(4, 30))
        d = float(rng.uniform(0.5, 12))
        if t + d >= seconds:
            break
        if rng.random() < 0.6:
            silences.append((round(t, 2), round(t + d, 2)))
        t += d
    batched = bool(rng.random() < 0.4)
    kw = dict(language=[None, "en", "zh", "de"][int(rng.integers(4))])
    if rng.random() < 0.5:
        kw["word_timestamps"] = True
    if rng.random() < 0.3:
        kw["multilingual"] = True
    if rng.random() < 0.3:
        kw["initial_prompt"] = ["hello world", " the test.", [260, 261]][int(rng.integers(3))]
    if rng.random() < 0.3:
        kw["hotwords"] = "whisper audio"
    if rng.random() < 0.3:
        kw["max_new_tokens"] = int(rng.integers(5, 120))
    if rng.random() < 0.3:
        kw["beam_size"] = int(rng.integers(1, 6))
    if rng.random() < 0.2:
        kw["suppress_tokens"] = [int(x) for x in rng.integers(0, 400, size=3)]
    if rng.random() < 0.15:
        kw["suppress_tokens"] = [None, [], [-1, 7]][int(rng.integers(3))]
    if rng.random() < 0.2:
        kw["suppress_blank"] = False
    if rng.random() < 0.2:
        kw["task"] = "translate"
    if rng.random() < 0.2:
        kw["patience"] = float(rng.choice([0.5, 2.0]))
        kw["length_penalty"] = float(rng.choice([0.0, 0.6, 2.0]))
    if rng.random() < 0.2:
        kw["repetition_penalty"] = 1.3
        kw["no_repeat_ngram_size"] = int(rng.integers(0, 4))
    if rng.random() < 0.15:
        kw["chunk_length"] = int(rng.choice([10, 20, 25]))
    if rng.random() < 0.2:
        kw["max_initial_timestamp"] = float(rng.choice([0.0, 0.5, 2.0]))
    if rng.random() < 0.15:
        kw["prepend_punctuations"] = "\"'([{"
        kw["append_punctuations"] = ".,!?)]}"
    if batched:
        if isinstance(kw.get("initial_prompt"), list):
            kw["initial_prompt"] = "hello"       # token-id prompts in the batched path are an extension here (the
                                                 # reference only takes a string there and raises TypeError otherwise)
        kw["batch_size"] = int(rng.integers(1, 5))
        kw["without_timestamps"] = bool(rng.random() < 0.5)
        mode = rng.random()
        if seconds < 30 and mode < 0.3:
            kw["vad_filter"] = False
        elif mode < 0.65:
            kw["vad_filter"] = True
            kw["vad_parameters"] = dict(min_silence_duration_ms=int(rng.integers(100, 1500)),
                                        speech_pad_ms=int(rng.integers(0, 500)))
        else:
            cuts = sorted(set([0.0, seconds] + [round(float(x), 2) for x in rng.uniform(0, seconds, size=4)]))
            clips = [dict(start=a, end=b) for a, b in zip(cuts[:-1], cuts[1:]) if 0.5 < b - a <= 30.0]
            if not clips:
                clips = [dict(start=0.0, end=min(seconds, 30.0))]
            kw["clip_timestamps"] = clips
    else:
        kw["without_timestamps"] = bool(rng.random() < 0.25)
        if rng.random() < 0.3:
            kw["prefix"] = "the model"
        if rng.random() < 0.4:
            kw["condition_on_previous_text"] = False
        if rng.random() < 0.4:
            kw["temperature"] = [[0.0], [0.0, 0.4, 0.8], 0.0, [0.2, 0.6]][int(rng.integers(4))]
        if rng.random() < 0.3:
            kw["best_of"] = int(rng.integers(1, 5))
        if rng.random() < 0.3:
            kw["no_speech_threshold"] = [None, 0.3, 0.9][int(rng.integers(3))]
        if rng.random() < 0.3:
            kw["log_prob_threshold"] = [None, -0.5, -1.5][int(rng.integers(3))]
        if rng.random() < 0.3:
            kw["compression_ratio_threshold"] = [None, 1.5, 3.0][int(rng.integers(3))]
        if rng.random() < 0.3:
            kw["prompt_reset_on_temperature"] = float(rng.choice([0.1, 0.5, 0.9]))
        if kw.get("word_timestamps") and rng.random() < 0.6:
            kw["hallucination_silence_threshold"] = float(rng.choice([0.5, 1.0, 2.0]))
        mode = rng.random()
        if mode < 0.25:
            kw["vad_filter"] = True
            kw["vad_parameters"] = dict(min_silence_duration_ms=int(rng.integers(100, 2500)))
        elif mode < 0.5:
            pts = sorted(round(float(x), 1) for x in rng.uniform(0, seconds, size=int(rng.integers(1, 5))))
            kw["clip_timestamps"] = pts if rng.random() < 0.5 else ",".join(str(p) for p in pts)
        if rng.random() < 0.2:
            kw["language_detection_segments"] = 2
            kw["language_detection_threshold"] = 0.8
    return dict(kind="batched" if batched else "sequential", audio=(1000 + seed, round(seconds, 2), tuple(silences)),
                kwargs=kw)


def run_reference(fw, cfg, hf_tok, case):
    from gen_golden_host import jsonable, make_reference_model
    from oracle import host_scenarios as hs
    model = make_reference_model(fw, cfg, hf_tok)
    audio = hs.synth_audio(*case["audio"])
    kwargs = json.loads(json.dumps(case["kwargs"]))
    if case["kind"] == "sequential":
        segments, info = model.transcribe(audio, **kwargs)
    else:
        segments, info = fw.BatchedInferencePipeline(model).transcribe(audio, **kwargs)
    segments = [jsonable(s) for s in segments]
    return dict(segments=segments, language=info.language, language_probability=float(info.language_probability),
                duration_after_vad=info.duration_after_vad, calls=jsonable(model.model.calls))


def run_mine(cfg, hf_tok, case):
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    from oracle import host_scenarios as hs
    from test_host_golden import _plain, make_model
    model = make_model(cfg, hf_tok)
    audio = hs.synth_audio(*case["audio"])
    kwargs = json.loads(json.dumps(case["kwargs"]))
    if kwargs.get("vad_filter"):
        kwargs["vad_speech_probs"] = hs.speech_probs(np.pad(audio, (0, 512 - audio.shape[0] % 512)))
    if case["kind"] == "sequential":
        segments, info = model.transcribe(audio, **kwargs)
    else:
        segments, info = BatchedInferencePipeline(model).transcribe(audio, **kwargs)
    segments = [_plain(s) for s in segments]
    return dict(segments=segments, language=info.language, language_probability=float(info.language_probability),
                duration_after_vad=info.duration_after_vad, calls=_plain(model.model.calls))


def fuzz_units(hf_tok, cfg, n_cases, seed):
    """random token sequences (raw bytes incl. partial UTF-8, merges, special tokens, timestamps): decode,
    decode_with_timestamps, split_to_word_tokens and merge_punctuations of both implementations must agree"""
    import faster_whisper.tokenizer as ref_tok
    from faster_whisper.transcribe import merge_punctuations as ref_merge
    from faster_whisper_amd import words as my_words
    from faster_whisper_amd.transcribe import Tokenizer
    rng = np.random.default_rng(seed)
    sample = hf_tok.encode(" hello world, 世界 café (test) \"a\" - x!", add_special_tokens=False).ids
    pre, app = "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、"
    bad = 0
    for trial in range(n_cases):
        lang = ("en", "zh", "de")[trial % 3]
        ref = ref_tok.Tokenizer(hf_tok, True, task="transcribe", language=lang)
        mine = Tokenizer(hf_tok, cfg, True, task="transcribe", language=lang)
        ids = []
        for _ in range(int(rng.integers(0, 25))):
            r = rng.random()
            if r < 0.55:
                ids.append(int(rng.integers(0, 400)))
            elif r < 0.8:
                ids.append(int(rng.choice(sample)))
            elif r < 0.9:
                ids.append(int(cfg.timestamp_begin + rng.integers(0, 1500)))
            else:
                ids.append(int(rng.integers(cfg.eot, cfg.timestamp_begin)))
        ids.append(cfg.eot)
        a, b = ref.split_to_word_tokens(list(ids)), mine.split_to_word_tokens(list(ids))
        al = [dict(word=w, tokens=list(t)) for w, t in zip(*a)]
        bl = [dict(word=w, tokens=list(t)) for w, t in zip(*b)]
        ref_merge(al, pre, app)
        my_words.merge_punctuations(bl, pre, app)
        ok = (a == b and al == bl and ref.decode(ids) == mine.decode(ids)
              and ref.decode_with_timestamps(ids) == mine.decode_with_timestamps(ids))
        if not ok:
            bad += 1
            if bad <= 3:
                print(f"unit mismatch ({lang}): ids={ids}\n  reference {a}\n  here      {b}")
    return bad


def fuzz_vad(n_cases, seed):
    """random probability tracks x random VadOptions: speech spans, chunk collection and the timestamp map of both
    implementations must agree exactly"""
    import faster_whisper.vad as rv
    from faster_whisper_amd import vad as mv
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(n_cases):
        n = int(rng.integers(1, 400))
        kind = int(rng.integers(4))
        if kind == 0:
            p = rng.random(n)
        elif kind == 1:
            p = np.clip(0.5 + np.cumsum(rng.normal(0, 0.15, n)) * 0.4, 0, 1)
        elif kind == 2:
            p = (rng.random(n) < 0.5).astype(float) * rng.uniform(0.4, 1.0) + 0.01
        else:
            p = np.full(n, 0.02)
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, n))
                p[a:min(n, a + int(rng.integers(1, 80)))] = rng.uniform(0.3, 1.0)
        n_audio = max(1, n * 512 - int(rng.integers(1, 512)) if rng.random() < 0.7 else (n - 1) * 512 if n > 1 else 300)
        p = np.resize(p, n_audio // 512 + 1)          # one probability per window of the padded audio
        opts = dict(threshold=float(rng.choice([0.3, 0.5, 0.7])))
        if rng.random() < 0.4:
            opts["neg_threshold"] = float(rng.choice([0.05, 0.2, 0.35]))
        if rng.random() < 0.5:
            opts["min_speech_duration_ms"] = int(rng.choice([0, 100, 250, 1000]))
        if rng.random() < 0.5:
            opts["max_speech_duration_s"] = float(rng.choice([1.0, 3.0, 8.0, 30.0]))
        if rng.random() < 0.7:
            opts["min_silence_duration_ms"] = int(rng.choice([0, 50, 100, 500, 2000]))
        if rng.random() < 0.7:
            opts["speech_pad_ms"] = int(rng.choice([0, 30, 100, 400]))
        rv.get_vad_model = lambda p=p: (lambda padded: p)
        audio = np.zeros(n_audio, np.float32)
        a = rv.get_speech_timestamps(audio, rv.VadOptions(**opts))
        b = mv.get_speech_timestamps(audio, mv.VadOptions(**opts), speech_probs=p)
        ok = a == b
        if ok and a:
            idx = np.arange(n_audio, dtype=np.float32)
            for md in (1.0, 5.0, float("inf")):
                ca, ma = rv.collect_chunks(idx, [dict(s) for s in a], max_duration=md)
                cb, mb = mv.collect_chunks(idx, [dict(s) for s in b], max_duration=md)
                ok = ok and ma == mb and len(ca) == len(cb) and all(np.array_equal(x, y) for x, y in zip(ca, cb))
            ta, tb = rv.SpeechTimestampsMap(a, 16000), mv.SpeechTimestampsMap(b, 16000)
            for q in rng.uniform(0, max(0.01, sum(s["end"] - s["start"] for s in a) / 16000), size=5):
                ok = ok and ta.get_original_time(q) == tb.get_original_time(q) \
                    and ta.get_original_time(q, is_end=True) == tb.get_original_time(q, is_end=True) \
                    and ta.get_chunk_index(q) == tb.get_chunk_index(q)
        if not ok:
            bad += 1
            if bad <= 3:
                print(f"vad mismatch: {opts} n_audio={n_audio}\n  reference {a[:4]}\n  here      {b[:4]}")
    return bad


def fuzz_logmel(n_cases, seed):
    """random lengths / amplitudes: oracle/logmel.py must stay BIT-identical to the reference's FeatureExtractor"""
    import faster_whisper.feature_extractor as ref_fe
    from oracle import logmel as olm
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(n_cases):
        n_mels = (80, 128)[trial % 2]
        fe = ref_fe.FeatureExtractor(feature_size=n_mels)
        length = int(rng.choice([0, 1, 159, 160, 161, 399, 400, 401, 1000, 16000, 48000, int(rng.integers(1, 200000))]))
        x = (rng.standard_normal(length) * float(rng.choice([1e-4, 0.01, 0.3, 1.0]))).astype(np.float32)
        if trial % 7 == 0:
            x[:] = 0
        a, b = fe(x), olm.log_mel_full(x, n_mels)
        if a.shape != b.shape or not np.array_equal(a, b):
            bad += 1
            if bad <= 3:
                print(f"log-mel mismatch: n_mels={n_mels} length={length} shapes {a.shape} {b.shape}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logmel", type=int, default=0, help="also fuzz N random waveforms through the log-mel oracle")
    ap.add_argument("--vad", type=int, default=0, help="also fuzz N random probability tracks through the VAD logic")
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--units", type=int, default=0, help="also fuzz N random token sequences through the tokenizer")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_host as gg
    gg.install_stubs()
    import faster_whisper as fw
    import faster_whisper.feature_extractor  # noqa: F401
    import faster_whisper.tokenizer as ref_tok
    import faster_whisper.vad as ref_vad
    from faster_whisper_amd import get_config
    from oracle import host_scenarios as hs
    from oracle import micro_tokenizer
    from test_host_golden import _close
    ref_tok._LANGUAGE_CODES = ("en", "zh", "de", "es")
    ref_vad.get_vad_model = lambda: hs.speech_probs
    cfg = get_config("micro")
    hf_tok = micro_tokenizer.build()
    bad = 0
    stats = dict(segments=0, words=0, generate=0, align=0, errors_equal=0)
    for seed in range(args.start, args.start + args.seeds):
        case = random_case(seed)
        try:
            want = run_reference(fw, cfg, hf_tok, case)
            ref_err = None
        except Exception as e:      # the reference itself rejects some argument combinations: so must we
            want, ref_err = None, type(e).__name__
        try:
            got = run_mine(cfg, hf_tok, case)
            my_err = None
        except Exception as e:
            got, my_err = None, type(e).__name__
        if ref_err or my_err:
            if ref_err != my_err:
                bad += 1
                print(f"seed {seed}: reference raised {ref_err}, this repository {my_err}: {case}")
            else:
                stats["errors_equal"] += 1
            continue
        try:
            _close(got, want, f"seed{seed}")
        except AssertionError as e:
            bad += 1
            print(f"seed {seed} MISMATCH {str(e)[:400]}\n  case: {case}")
            continue
        stats["segments"] += len(want["segments"])
        stats["words"] += sum(len(s["words"] or []) for s in want["segments"])
        stats["generate"] += sum(1 for c in want["calls"] if c[0] == "generate")
        stats["align"] += sum(1 for c in want["calls"] if c[0] == "align")
    unit_bad = fuzz_units(hf_tok, cfg, args.units, args.start) if args.units else 0
    vad_bad = fuzz_vad(args.vad, args.start) if args.vad else 0
    ref_vad.get_vad_model = lambda: hs.speech_probs
    mel_bad = fuzz_logmel(args.logmel, args.start) if args.logmel else 0
    print(json.dumps(dict(seeds=args.seeds, mismatches=bad, units=args.units, unit_mismatches=unit_bad, vad=args.vad,
                          vad_mismatches=vad_bad, logmel=args.logmel, logmel_mismatches=mel_bad, **stats)))
    return 1 if (bad or unit_bad or vad_bad or mel_bad) else 0


if __name__ == "__main__":
    sys.exit(main())
