"""Generates tests/golden/logmel_*.npz by running the REFERENCE implementation
(/root/reference/faster_whisper/feature_extractor.py, loaded standalone because
faster_whisper/__init__.py imports PyAV which is not installed here).

Run in the build container only (the GPU box has no /root/reference):
    python oracle/gen_golden.py
The fixtures it writes are committed; tests never import the reference.
"""
import importlib.util
import os
import wave

import numpy as np

REF = "/root/reference/faster_whisper/feature_extractor.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_feature_extractor", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_pad_or_trim(a, length=3000):
    # audio.py:111-123 semantics (trim, or right-pad with zeros)
    if a.shape[-1] > length:
        a = a[..., :length]
    if a.shape[-1] < length:
        a = np.pad(a, [(0, 0), (0, length - a.shape[-1])])
    return a


def synth_cases():
    """name -> list of float32 chunks (ragged). Deterministic: tests regenerate the same inputs."""
    rng = np.random.default_rng(20250921)
    t = np.arange(480000) / 16000.0
    cases = {}
    # bench-style audio: noise + three partials
    bench = (0.1 * rng.standard_normal(480000) + 0.05 * np.sin(2 * np.pi * 220 * t)
             + 0.05 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 880 * t)).astype(np.float32)
    cases["bench30s"] = [bench]
    cases["ragged"] = [
        (0.3 * rng.standard_normal(n)).astype(np.float32) for n in (0, 1, 41, 159, 160, 161, 1000, 16000, 123457)
    ]
    cases["silence_and_click"] = [np.zeros(32000, np.float32),
                                  np.r_[np.zeros(8000), 1.0, np.zeros(7999)].astype(np.float32)]
    cases["tone"] = [(0.5 * np.sin(2 * np.pi * 1000.0 * t[:160000])).astype(np.float32)]
    return cases


def speech_case():
    """the only reference audio fixture decodable with the stdlib: PCM16 stereo 16 kHz (5 s)"""
    wav = "/root/reference/tests/data/stereo_diarization.wav"
    with wave.open(wav, "rb") as w:
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).reshape(-1, w.getnchannels())
    return (raw.astype(np.float32).mean(axis=1) / 32768.0).astype(np.float32)


def main():
    ref = load_reference()
    os.makedirs(OUT, exist_ok=True)
    speech = speech_case()
    # the speech input itself is small (80000 samples): commit it as int16-exact float16-free npz
    np.savez_compressed(os.path.join(OUT, "speech_pcm.npz"), pcm=speech)
    for n_mels in (80, 128):
        fe = ref.FeatureExtractor(feature_size=n_mels)
        np.save(os.path.join(OUT, f"mel_filters_{n_mels}.npy"), fe.mel_filters)
        cases = synth_cases()
        cases["speech"] = [speech]
        for name, chunks in cases.items():
            feats = np.stack([ref_pad_or_trim(fe(c)[..., :-1]) for c in chunks]).astype(np.float32)
            full = fe(chunks[-1]).astype(np.float32)  # whole-waveform variant (sequential path, transcribe.py:916)
            nz = int(max(len(c) for c in chunks) // 160) + 2
            keep = min(3000, nz)
            idx = np.arange(keep) if keep <= 1100 else np.unique(
                np.r_[0:64, np.linspace(0, keep - 1, 384).astype(int), keep - 64:keep])
            fidx = idx[idx < full.shape[-1]]
            np.savez_compressed(
                os.path.join(OUT, f"logmel_{name}_{n_mels}.npz"),
                n_samples=np.array([len(c) for c in chunks]), frame_idx=idx, feats=feats[..., idx],
                rest_absmax=np.array([float(np.abs(np.delete(feats, idx, axis=-1)).max()) if keep < 3000 else -1.0]),
                full_idx=fidx, full_last=full[..., fidx], full_frames=np.array([full.shape[-1]]))
    print("golden fixtures written to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
