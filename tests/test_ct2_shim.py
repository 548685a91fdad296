"""The drop-in claim at the Python level: with `faster_whisper_amd.ct2_shim` registered as `ctranslate2`, the
UNMODIFIED reference package imports and binds this repository's backend classes.  Needs the reference checkout
(build container); runs in a subprocess so that the stub modules do not leak into the test session.  No GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CODE = r"""
import sys
sys.path.insert(0, {root!r})
import faster_whisper_amd.ct2_shim as shim
shim.install(stub_av=True)
sys.path.insert(0, {ref!r})
import ctranslate2, faster_whisper
from faster_whisper_amd import backend
assert ctranslate2.models.Whisper is backend.Whisper and ctranslate2.StorageView is backend.StorageView
assert faster_whisper.transcribe.ctranslate2 is ctranslate2
import inspect
sig = inspect.signature(backend.Whisper.__init__)
for name in ("device", "device_index", "compute_type", "intra_threads", "inter_threads", "files"):
    assert name in sig.parameters, name                     # the keyword arguments of transcribe.py:689-698
gen = inspect.signature(backend.Whisper.generate)
for name in ("beam_size", "patience", "num_hypotheses", "length_penalty", "repetition_penalty", "no_repeat_ngram_size",
             "max_length", "return_scores", "return_no_speech_prob", "max_initial_timestamp_index", "suppress_blank",
             "suppress_tokens", "sampling_topk", "sampling_temperature"):
    assert name in gen.parameters, name                     # transcribe.py:222-236, :1433-1459
assert "median_filter_width" in inspect.signature(backend.Whisper.align).parameters
assert ctranslate2.get_supported_compute_types("cuda") >= {{"float16", "int8_float16"}}
print("ok")
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "faster_whisper")), reason="reference checkout not on this box")
def test_reference_package_imports_on_the_shim():
    r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, ref=REF)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
