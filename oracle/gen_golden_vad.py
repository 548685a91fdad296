"""Generates tests/golden/vad_speech_probs.npy: the Silero VAD v6 speech probabilities of the reference's
speech fixture (tests/golden/speech_pcm.npz = /root/reference/tests/data/stereo_diarization.wav, mono) between
2 s / 1.94 s of digital silence, computed by oracle/silero.py (numpy restatement of the ONNX graph) from the
reference's asset /root/reference/faster_whisper/assets/silero_vad_v6.onnx.

Build container only.  PARITY UNPINNED: onnxruntime is absent, so these are NOT outputs of the reference's own
run; they pin this repository's implementations (numpy restatement and C++) to each other across rounds and
carry the plausibility check (speech high, silence low).
    python oracle/gen_golden_vad.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from faster_whisper_amd import onnx_lite  # noqa: E402
from oracle import silero  # noqa: E402

ONNX = "/root/reference/faster_whisper/assets/silero_vad_v6.onnx"


def main():
    _, inits, _, _ = onnx_lite.load(ONNX)
    speech = np.load(os.path.join(ROOT, "tests", "golden", "speech_pcm.npz"))["pcm"].astype(np.float32)
    audio = np.concatenate([np.zeros(32000, np.float32), speech, np.zeros(31000, np.float32)])
    padded = np.pad(audio, (0, 512 - len(audio) % 512))
    probs, _, _ = silero.forward(inits, silero.frame_windows(padded))
    out = os.path.join(ROOT, "tests", "golden", "vad_speech_probs.npy")
    np.save(out, probs.astype(np.float32))
    print(out, probs.shape, "silence", float(probs[:55].max()), "speech mean", float(probs[70:220].mean()))


if __name__ == "__main__":
    main()
