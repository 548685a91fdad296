"""Generates tests/golden/host_*.json by running the REFERENCE's host code
(/root/reference/faster_whisper: transcribe.py, tokenizer.py, vad.py, feature_extractor.py) on the inputs of
oracle/host_scenarios.py with oracle/scripted_backend.py in the place of `ctranslate2.models.Whisper`.

`ctranslate2`, `av` and `onnxruntime` are not installed: tiny stub modules stand in for the imports (only
`StorageView.from_array` is ever called).  Run in the build container only (the GPU box has no /root/reference):
    python oracle/gen_golden_host.py
The fixtures are committed; tests never import the reference.
"""
import dataclasses
import json
import logging
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


class _StorageView:
    def __init__(self, a):
        self.array = a

    @classmethod
    def from_array(cls, a):
        return cls(a)

    def __array__(self, dtype=None, copy=None):
        return self.array if dtype is None else self.array.astype(dtype)


def install_stubs():
    ct2 = types.ModuleType("ctranslate2")
    ct2.StorageView = _StorageView
    ct2.models = types.ModuleType("ctranslate2.models")
    ct2.models.Whisper = object
    ct2.models.WhisperGenerationResult = object
    ct2.get_supported_compute_types = lambda *a, **k: ["float16"]
    sys.modules["ctranslate2"] = ct2
    sys.modules["ctranslate2.models"] = ct2.models

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Any(name)

        def __call__(self, *a, **k):
            return _Any("call")
    for name in ("av", "av.audio", "av.audio.resampler", "av.error"):
        sys.modules[name] = _Any(name)
    sys.path.insert(0, "/root/reference")


def jsonable(o):
    if dataclasses.is_dataclass(o):
        return {k: jsonable(v) for k, v in dataclasses.asdict(o).items()}
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    return o


def make_reference_model(fw, cfg, hf_tok):
    from oracle.scripted_backend import ScriptedBackend
    m = fw.WhisperModel.__new__(fw.WhisperModel)
    m.logger = logging.getLogger("ref")
    m.model = ScriptedBackend(cfg, hf_tok)
    m.hf_tokenizer = hf_tok
    m.feat_kwargs = {}
    m.feature_extractor = fw.feature_extractor.FeatureExtractor(feature_size=cfg.n_mels)
    m.input_stride = 2
    m.num_samples_per_token = m.feature_extractor.hop_length * m.input_stride
    m.frames_per_second = m.feature_extractor.sampling_rate // m.feature_extractor.hop_length
    m.tokens_per_second = m.feature_extractor.sampling_rate // m.num_samples_per_token
    m.time_precision = 0.02
    m.max_length = 448
    return m


def run_scenarios(fw, cfg, hf_tok):
    import faster_whisper.vad as ref_vad
    from oracle import host_scenarios as hs
    ref_vad.get_vad_model = lambda: hs.speech_probs
    # the reference's language list is the real 99/100 codes; the micro vocabulary has the first four
    out = {}
    for name, sc in hs.SCENARIOS.items():
        model = make_reference_model(fw, cfg, hf_tok)
        audio = hs.synth_audio(*sc["audio"])
        kwargs = json.loads(json.dumps(sc["kwargs"]))     # deep copy (the reference mutates vad dicts)
        if sc["kind"] == "sequential":
            segments, info = model.transcribe(audio, **kwargs)
        else:
            pipe = fw.BatchedInferencePipeline(model)
            segments, info = pipe.transcribe(audio, **kwargs)
        segments = [jsonable(s) for s in segments]
        out[name] = dict(
            segments=segments,
            info=dict(language=info.language, language_probability=float(info.language_probability),
                      duration=info.duration, duration_after_vad=info.duration_after_vad,
                      all_language_probs=jsonable(info.all_language_probs)),
            calls=jsonable(model.model.calls))
        print(f"{name}: {len(segments)} segments, {len(model.model.calls)} backend calls, "
              f"language {info.language}")
    return out


def run_units(fw, cfg, hf_tok):
    import faster_whisper.vad as ref_vad
    from faster_whisper.tokenizer import Tokenizer as RefTokenizer
    from faster_whisper.transcribe import merge_punctuations
    from oracle import host_scenarios as hs
    out = {"vad": {}, "chunks": {}, "ts_map": {}, "split": {}, "merge": {}}
    for tname, (n_audio, probs) in hs.vad_prob_tracks().items():
        for cname, opts in hs.VAD_CASES.items():
            ref_vad.get_vad_model = lambda probs=probs: (lambda padded: probs)
            audio = np.zeros(n_audio, dtype=np.float32)
            spans = ref_vad.get_speech_timestamps(audio, ref_vad.VadOptions(**opts))
            key = f"{tname}/{cname}"
            out["vad"][key] = jsonable(spans)
            audio_idx = np.arange(n_audio, dtype=np.float32)
            for md in (30.0, 7.5):
                chunks, meta = ref_vad.collect_chunks(audio_idx, [dict(s) for s in spans], max_duration=md)
                out["chunks"][f"{key}/{md}"] = dict(
                    lens=[int(len(c)) for c in chunks], first=[float(c[0]) if len(c) else None for c in chunks],
                    last=[float(c[-1]) if len(c) else None for c in chunks], meta=jsonable(meta))
            if spans:
                m = ref_vad.SpeechTimestampsMap(spans, 16000)
                total = sum(s["end"] - s["start"] for s in spans) / 16000
                qs = [0.0, 0.5, total / 3, total / 2, total * 0.9, total]
                out["ts_map"][key] = dict(
                    queries=qs, plain=[m.get_original_time(q) for q in qs],
                    ends=[m.get_original_time(q, is_end=True) for q in qs],
                    index=[m.get_chunk_index(q) for q in qs])
    for lang, texts in hs.SPLIT_TEXTS.items():
        tok = RefTokenizer(hf_tok, True, task="transcribe", language=lang)
        for text in texts:
            ids = tok.encode(text) + [tok.timestamp_begin + 10] + tok.encode(" and") + [tok.eot]
            words, groups = tok.split_to_word_tokens(ids)
            out["split"][f"{lang}|{text}"] = dict(ids=ids, words=words, groups=groups,
                                                 decoded=tok.decode_with_timestamps(ids))
            alignment = [dict(word=w, tokens=list(g)) for w, g in zip(words, groups)]
            merge_punctuations(alignment, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、")
            out["merge"][f"{lang}|{text}"] = alignment
    return out


def api_surface(fw):
    """the reference's public call surface for this path: parameter names, order and defaults of the entry points, the
    field lists of the result dataclasses, the package exports (faster_whisper/__init__.py, transcribe.py:66-108,
    :589-698, :720-780, :254-330; vad.py:14-43; audio.py:20-47)"""
    import inspect
    from faster_whisper import audio as ref_audio
    from faster_whisper import transcribe as ref_tr
    from faster_whisper import vad as ref_vad

    def sig(f):
        out = []
        for name, p in inspect.signature(f).parameters.items():
            if name == "self":
                continue
            d = None if p.default is inspect.Parameter.empty else repr(p.default)
            out.append([name, {"VAR_KEYWORD": "**", "VAR_POSITIONAL": "*"}.get(p.kind.name, ""), d])
        return out

    def fields(c):
        return [f.name for f in dataclasses.fields(c)]
    return {
        "exports": sorted(fw.__all__),
        "signatures": {
            "WhisperModel.__init__": sig(ref_tr.WhisperModel.__init__),
            "WhisperModel.transcribe": sig(ref_tr.WhisperModel.transcribe),
            "WhisperModel.detect_language": sig(ref_tr.WhisperModel.detect_language),
            "BatchedInferencePipeline.__init__": sig(ref_tr.BatchedInferencePipeline.__init__),
            "BatchedInferencePipeline.transcribe": sig(ref_tr.BatchedInferencePipeline.transcribe),
            "decode_audio": sig(ref_audio.decode_audio),
            "pad_or_trim": sig(ref_audio.pad_or_trim),
            "get_speech_timestamps": sig(ref_vad.get_speech_timestamps),
            "collect_chunks": sig(ref_vad.collect_chunks),
        },
        "dataclasses": {
            "Word": fields(ref_tr.Word), "Segment": fields(ref_tr.Segment),
            "TranscriptionOptions": fields(ref_tr.TranscriptionOptions),
            "TranscriptionInfo": fields(ref_tr.TranscriptionInfo), "VadOptions": fields(ref_vad.VadOptions),
        },
        "vad_defaults": dataclasses.asdict(ref_vad.VadOptions()),
    }


def main():
    install_stubs()
    import faster_whisper as fw
    if "--api-only" in sys.argv:
        with open(os.path.join(OUT, "host_api.json"), "w") as f:
            json.dump(api_surface(fw), f, indent=1)
        print("wrote host_api.json")
        return
    import faster_whisper.feature_extractor  # noqa: F401
    import faster_whisper.tokenizer as ref_tok
    from faster_whisper_amd import get_config
    from oracle import micro_tokenizer
    # the micro vocabulary only has four language tokens
    ref_tok._LANGUAGE_CODES = ("en", "zh", "de", "es")
    cfg = get_config("micro")
    hf_tok = micro_tokenizer.build()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "host_scenarios.json"), "w") as f:
        json.dump(run_scenarios(fw, cfg, hf_tok), f, indent=0, ensure_ascii=False)
    with open(os.path.join(OUT, "host_units.json"), "w") as f:
        json.dump(run_units(fw, cfg, hf_tok), f, indent=0, ensure_ascii=False)
    with open(os.path.join(OUT, "host_api.json"), "w") as f:
        json.dump(api_surface(fw), f, indent=1)
    print("wrote host_scenarios.json, host_units.json, host_api.json")


if __name__ == "__main__":
    main()
