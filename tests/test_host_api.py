"""Drop-in call surface: every parameter of the reference's entry points for this path exists here under the same name,
at the same position, with the same default (extras only after the reference's last parameter), the result dataclasses
have the reference's fields in the reference's order, and the package exports the reference's names.
tests/golden/host_api.json is read off the REFERENCE package by oracle/gen_golden_host.py (build container); this
mirrors and widens the reference's own tests/test_transcribe.py:237-244 (test_transcribe_signature).  No GPU."""
import ast
import dataclasses
import inspect
import json
import os

import pytest

import faster_whisper_amd as fwa
from faster_whisper_amd import audio as f_audio
from faster_whisper_amd import transcribe as f_tr
from faster_whisper_amd import vad as f_vad

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_api.json")) as _f:
    API = json.load(_f)

HERE = {
    "WhisperModel.__init__": f_tr.WhisperModel.__init__,
    "WhisperModel.transcribe": f_tr.WhisperModel.transcribe,
    "WhisperModel.detect_language": f_tr.WhisperModel.detect_language,
    "BatchedInferencePipeline.__init__": f_tr.BatchedInferencePipeline.__init__,
    "BatchedInferencePipeline.transcribe": f_tr.BatchedInferencePipeline.transcribe,
    "decode_audio": f_audio.decode_audio,
    "pad_or_trim": f_tr.pad_or_trim,                  # lives beside its only caller here
    "get_speech_timestamps": f_vad.get_speech_timestamps,
    "collect_chunks": f_vad.collect_chunks,
}


def _norm(d):
    """a default's repr, with tuple and list literals treated alike (an immutable default here for a list there)"""
    try:
        v = ast.literal_eval(d)
    except (ValueError, SyntaxError, TypeError):
        return d
    return repr(list(v)) if isinstance(v, (tuple, list)) else repr(v)


def _sig(f):
    out = []
    for name, p in inspect.signature(f).parameters.items():
        if name == "self":
            continue
        d = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append([name, {"VAR_KEYWORD": "**", "VAR_POSITIONAL": "*"}.get(p.kind.name, ""), d])
    return out


@pytest.mark.parametrize("name", sorted(API["signatures"]))
def test_signature_is_a_positional_superset(name):
    want, got = API["signatures"][name], _sig(HERE[name])
    var_kw = [p for p in want if p[1] == "**"]
    want = [p for p in want if p[1] != "**"]
    assert len(got) >= len(want), (name, [p[0] for p in got])
    for i, (w, g) in enumerate(zip(want, got)):
        assert g[0] == w[0], f"{name}: parameter {i} is `{g[0]}`, the reference has `{w[0]}`"
        assert _norm(g[2]) == _norm(w[2]), f"{name}: `{w[0]}` defaults to {g[2]}, the reference to {w[2]}"
    extras = got[len(want):]
    # what this repository adds comes after the reference's parameters and is optional
    assert all(p[2] is not None or p[1] == "**" for p in extras), (name, extras)
    if var_kw:
        assert any(p[1] == "**" for p in got), f"{name}: the reference accepts **{var_kw[0][0]}"


def test_batched_and_sequential_transcribe_take_the_same_arguments():
    """the reference's own test_transcribe_signature, on the reference's parameter set"""
    ref_seq = {p[0] for p in API["signatures"]["WhisperModel.transcribe"]}
    ref_bat = {p[0] for p in API["signatures"]["BatchedInferencePipeline.transcribe"]} - {"batch_size"}
    assert ref_seq == ref_bat                                   # the fixture says what the reference's test says
    seq = set(inspect.signature(f_tr.WhisperModel.transcribe).parameters)
    bat = set(inspect.signature(f_tr.BatchedInferencePipeline.transcribe).parameters)
    assert ref_seq <= seq and ref_bat <= bat


@pytest.mark.parametrize("name", sorted(API["dataclasses"]))
def test_dataclass_fields(name):
    cls = getattr(f_vad if name == "VadOptions" else f_tr, name)
    assert [f.name for f in dataclasses.fields(cls)] == API["dataclasses"][name]


def test_vad_defaults_and_exports():
    assert dataclasses.asdict(f_vad.VadOptions()) == API["vad_defaults"]
    # `download_model` (Hugging Face Hub fetch) is the one export outside this tier: SURVEY.md section 2 marks model
    # download out of scope (no network); WhisperModel takes a local CTranslate2 model directory
    missing = [n for n in API["exports"] if n != "download_model" and not hasattr(fwa, n)]
    assert not missing, missing
