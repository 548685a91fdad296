"""The drop-in boundary EXECUTED: the UNMODIFIED reference package (faster_whisper/transcribe.py: WhisperModel,
BatchedInferencePipeline, the sequential seek loop, word timestamps) runs on this repository's backend registered as
`ctranslate2` (faster_whisper_amd/ct2_shim.py), and every Segment / Word / TranscriptionInfo it yields is compared with
what this repository's own host code (faster_whisper_amd/transcribe.py) yields on the same engine.

    python profiles/run_reference_on_shim.py --ref <checkout of SYSTRAN/faster-whisper> [--backend engine|oracle]

The reference checkout is NOT part of this repository and does not exist on the GPU box: for the one scratch run
whose log is kept as profiles/r03_reference_on_shim.log it was copied next to the repository snapshot (untracked,
git-ignored) and removed afterwards.  --backend oracle swaps the CPU restatement in for the engine so that the
script itself can be checked in the build container (no GPU there).
Model: the `micro` geometry with seeded synthetic weights in a model directory (fwamd_config.json +
weights.safetensors + tokenizer.json from oracle/micro_tokenizer.py); no Whisper checkpoint exists offline.
"""
import argparse
import dataclasses
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def seg_tuple(s):
    words = None
    if s.words is not None:
        words = [(w.word, round(w.start, 3), round(w.end, 3), round(float(w.probability), 6)) for w in s.words]
    return dict(id=s.id, seek=s.seek, start=round(s.start, 3), end=round(s.end, 3), text=s.text, tokens=list(s.tokens),
                avg_logprob=round(float(s.avg_logprob), 6), no_speech_prob=round(float(s.no_speech_prob), 6),
                temperature=s.temperature, compression_ratio=round(float(s.compression_ratio), 6), words=words)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.path.join(ROOT, "_ref_scratch"))
    ap.add_argument("--backend", default="engine", choices=["engine", "oracle"])
    args = ap.parse_args()
    assert os.path.isdir(os.path.join(args.ref, "faster_whisper")), f"no reference checkout at {args.ref}"

    import faster_whisper_amd.ct2_shim as shim
    from faster_whisper_amd import get_config, synthetic_weights
    from faster_whisper_amd.backend import save_model_dir
    from conftest import bench_audio
    from oracle import micro_tokenizer

    cfg = get_config("micro")
    weights = synthetic_weights(cfg, seed=7)
    mdir = tempfile.mkdtemp(prefix="fwamd_micro_")
    save_model_dir(mdir, cfg, weights)
    micro_tokenizer.build().save(os.path.join(mdir, "tokenizer.json"))

    if args.backend == "oracle":
        from oracle.oracle_backend import OracleBackend
        from faster_whisper_amd.backend import load_model_dir

        class _OracleAsCt2(OracleBackend):           # the constructor signature of transcribe.py:689-698
            def __init__(self, model_path, device="auto", device_index=0, compute_type="default", intra_threads=0,
                         inter_threads=1, files=None, **kw):
                c, w = load_model_dir(model_path)
                super().__init__(c, w, emulate_fp16=True)
        shim.models.Whisper = _OracleAsCt2
    shim.install(stub_av=True)
    sys.path.insert(0, args.ref)
    import faster_whisper as ref                       # the reference package, unmodified
    import faster_whisper_amd.transcribe as ours
    print("reference package:", os.path.dirname(ref.__file__), "version", ref.__version__)
    assert ref.transcribe.ctranslate2 is sys.modules["ctranslate2"]

    dev = "cuda" if args.backend == "engine" else "cpu"
    kw_model = dict(device=dev, compute_type="float16")
    rm = ref.WhisperModel(mdir, **kw_model)            # reference constructor -> shim -> fw_model_create
    if args.backend == "engine":
        om = ours.WhisperModel(mdir, **kw_model)
    else:                                              # our host code on the very same backend object
        import logging
        om = ours.WhisperModel.__new__(ours.WhisperModel)
        om.logger = logging.getLogger("ours")
        om.model, om.hf_tokenizer = rm.model, rm.hf_tokenizer
        om.feat_kwargs = {}
        om.feature_extractor = ours.FeatureExtractor(feature_size=cfg.n_mels, backend=rm.model)
        om.input_stride, om.time_precision, om.max_length = 2, 0.02, 448
        om.num_samples_per_token = 320
        om.frames_per_second, om.tokens_per_second = 100, 50
    print("backend:", type(rm.model).__module__, type(rm.model).__name__, "| multilingual:", rm.model.is_multilingual)

    audio = np.concatenate([bench_audio(480000, seed=60 + i) for i in range(3)])[:int(75.0 * 16000)]
    sup = [1, 2, 3]
    n_checked = 0
    exact = args.backend == "oracle"      # same backend object AND the same (numpy) log-mel on both sides

    def compare(what, a_res, b_res):
        nonlocal n_checked
        (a_segs, a_info), (b_segs, b_info) = a_res, b_res
        a_segs, b_segs = [seg_tuple(s) for s in a_segs], [seg_tuple(s) for s in b_segs]
        ia, ib = dataclasses.asdict(a_info), dataclasses.asdict(b_info)
        for k in ("transcription_options", "vad_options"):
            ia.pop(k, None), ib.pop(k, None)
        assert len(a_segs) > 0 and len(b_segs) > 0, (what, len(a_segs), len(b_segs))
        if exact:
            assert len(a_segs) == len(b_segs), (what, len(a_segs), len(b_segs))
            for x, y in zip(a_segs, b_segs):
                assert x == y, (what, x, y)
            same = len(a_segs)
        else:
            # engine run: the reference computes the log-mel in numpy on the host, this repository's front on the GPU
            # (<= 6e-5 apart), so floats agree to the parity tolerances and a numerically tied step may fork the text
            same = 0
            for x, y in zip(a_segs, b_segs):
                if (x["tokens"], x["seek"], x["text"], x["id"]) != (y["tokens"], y["seek"], y["text"], y["id"]):
                    break
                assert abs(x["start"] - y["start"]) <= 0.021 and abs(x["end"] - y["end"]) <= 0.021, (what, x, y)
                assert abs(x["avg_logprob"] - y["avg_logprob"]) < 2e-3 * max(1.0, abs(y["avg_logprob"])), (what, x, y)
                assert abs(x["no_speech_prob"] - y["no_speech_prob"]) < 2e-3, (what, x, y)
                if x["words"] is not None:
                    assert [w[0] for w in x["words"]] == [w[0] for w in y["words"]], (what, x, y)
                    for wa, wb in zip(x["words"], y["words"]):
                        assert abs(wa[1] - wb[1]) <= 0.045 and abs(wa[2] - wb[2]) <= 0.045 and abs(wa[3] - wb[3]) < 2e-3
                same += 1
            assert same >= max(1, (len(b_segs) + 1) // 2), (what, same, len(a_segs), len(b_segs))
        for k in ("language", "duration", "duration_after_vad"):
            assert ia[k] == ib[k], (what, k, ia[k], ib[k])
        assert abs(ia["language_probability"] - ib["language_probability"]) < (1e-6 if exact else 2e-3)
        n_checked += same
        print(f"{what}: {same} of {len(b_segs)} leading segments identical (reference host code vs this repository's host code); "
              f"language {ia['language']} p={ia['language_probability']:.4f}; first: {a_segs[0]['text'][:40]!r} "
              f"avg_logprob {a_segs[0]['avg_logprob']}")

    # ---- BatchedInferencePipeline.transcribe (transcribe.py:254-617), the hot path ----
    clips = [{"start": 0.0, "end": 30.0}, {"start": 30.0, "end": 52.5}, {"start": 52.5, "end": 75.0}]
    bkw = dict(language="en", beam_size=5, batch_size=2, clip_timestamps=clips, max_new_tokens=16, suppress_tokens=sup,
               without_timestamps=True, log_prob_threshold=None, no_speech_threshold=None)
    compare("batched beam 5", ref.BatchedInferencePipeline(rm).transcribe(audio, **bkw),
            ours.BatchedInferencePipeline(om).transcribe(audio, **bkw))
    bkw2 = dict(bkw, word_timestamps=True, without_timestamps=False, language=None, multilingual=False)
    compare("batched + language detection + word timestamps", ref.BatchedInferencePipeline(rm).transcribe(audio, **bkw2),
            ours.BatchedInferencePipeline(om).transcribe(audio, **bkw2))
    # ---- WhisperModel.transcribe (sequential path, transcribe.py:747-1022, :1103-1389) ----
    skw = dict(language="en", beam_size=2, temperature=0.0, word_timestamps=True, max_new_tokens=14,
               log_prob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None, suppress_tokens=sup)
    compare("sequential beam 2 + word timestamps", rm.transcribe(audio[:int(52 * 16000)], **skw),
            om.transcribe(audio[:int(52 * 16000)], **skw))
    print(f"OK: {n_checked} segments compared; backend = {args.backend}")


if __name__ == "__main__":
    main()
