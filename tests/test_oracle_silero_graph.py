"""oracle/silero.py (the HAND restatement of the reference's VAD network) against a GENERIC execution of the asset's own
graph (oracle/onnx_exec.py: the node list of faster_whisper/assets/silero_vad_v6.onnx evaluated operator by operator per
the ONNX specification on torch's conv1d / LSTM kernels).  onnxruntime — what vad.py:288-351 runs — is not installed
anywhere this suite runs, so this is the pin the restatement can have: the reference's model FILE under the operator
specification, derived twice independently.  No GPU.

  * with random weights, on the committed topology (tests/golden/silero_graph_nodes.json, made by
    tests/golden/make_silero_graph_golden.py): runs on every box;
  * with the real asset (build container): the topology fixture equals the asset's node list, and on the reference's
    speech fixture executor == restatement == the committed probabilities (tests/golden/vad_speech_probs.npy)."""
import json
import os

import numpy as np
import pytest

from faster_whisper_amd import onnx_lite
from oracle import onnx_exec, silero

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_ONNX = "/root/reference/faster_whisper/assets/silero_vad_v6.onnx"


def _topology():
    with open(os.path.join(GOLD, "silero_graph_nodes.json")) as f:
        doc = json.load(f)
    consts = {k: np.asarray(v["values"], dtype=v["dtype"]).reshape(v["shape"]) for k, v in doc["constants"].items()}
    return doc, consts


def _random_weights(shapes, seed):
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in shapes.items():
        scale = 0.05 if "basis" in name else (1.5 if name.startswith("decoder.") else 0.12)
        w[name] = (rng.standard_normal(shape) * scale).astype(np.float32)
    return w


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 7), (2, 40)])
def test_restatement_equals_graph_execution_random_weights(seed, n):
    doc, consts = _topology()
    weights = _random_weights(doc["weight_shapes"], seed)
    rng = np.random.default_rng(100 + seed)
    win = (rng.standard_normal((n, 576)) * 0.4).astype(np.float32)
    h0 = (rng.standard_normal((1, 1, 128)) * 0.5).astype(np.float32)
    c0 = (rng.standard_normal((1, 1, 128)) * 0.5).astype(np.float32)
    probs, hn, cn = onnx_exec.run(doc["nodes"], {**consts, **weights}, {"input": win, "h": h0, "c": c0}, doc["outputs"])
    ref, rh, rc = silero.forward(weights, win, h0.reshape(128), c0.reshape(128))
    assert probs.shape == (n,) and hn.shape == (1, 1, 128) and cn.shape == (1, 1, 128)
    assert 2e-3 < probs.std() or n == 1                                 # not a degenerate constant
    print(f"graph execution vs restatement, {n} windows: probs {np.abs(probs - ref).max():.2e}, "
          f"h {np.abs(hn.reshape(128) - rh).max():.2e}, c {np.abs(cn.reshape(128) - rc).max():.2e}")
    assert np.abs(probs - ref).max() < 2e-6
    assert np.abs(hn.reshape(128) - rh).max() < 2e-6 and np.abs(cn.reshape(128) - rc).max() < 2e-6


def test_state_hand_over_is_the_graphs():
    """two executions with hn / cn handed over == one (vad.py:338-347 carries the state across batches of 10 000 windows)"""
    doc, consts = _topology()
    weights = _random_weights(doc["weight_shapes"], 9)
    win = (np.random.default_rng(3).standard_normal((12, 576)) * 0.4).astype(np.float32)
    z = np.zeros((1, 1, 128), np.float32)
    inits = {**consts, **weights}
    whole, h1, c1 = onnx_exec.run(doc["nodes"], inits, {"input": win, "h": z, "c": z}, doc["outputs"])
    a, h, c = onnx_exec.run(doc["nodes"], inits, {"input": win[:5], "h": z, "c": z}, doc["outputs"])
    b, h, c = onnx_exec.run(doc["nodes"], inits, {"input": win[5:], "h": h, "c": c}, doc["outputs"])
    assert np.abs(np.concatenate([a, b]) - whole).max() < 1e-6 and np.abs(h - h1).max() < 1e-6 and np.abs(c - c1).max() < 1e-6


def test_executor_refuses_what_it_does_not_implement():
    with pytest.raises(NotImplementedError):
        onnx_exec.run([dict(op="Gemm", inputs=["x"], outputs=["y"], attrs={})], {}, {"x": np.zeros(2, np.float32)}, ["y"])
    with pytest.raises(KeyError):
        onnx_exec.run([dict(op="Relu", inputs=["nope"], outputs=["y"], attrs={})], {}, {}, ["y"])
    # reflect padding without edge duplication, Slice bounds clamped ("INT64_MAX = to the end")
    x = np.arange(6, dtype=np.float32).reshape(1, 6)
    y, = onnx_exec.run([dict(op="Pad", inputs=["x", "p"], outputs=["y"], attrs={"mode": "reflect"})],
                       {"p": np.array([0, 2, 0, 1])}, {"x": x}, ["y"])
    assert y.tolist() == [[2, 1, 0, 1, 2, 3, 4, 5, 4]]
    s, = onnx_exec.run([dict(op="Slice", inputs=["x", "s", "e", "a", "t"], outputs=["y"], attrs={})],
                       {"s": np.array([2]), "e": np.array([onnx_exec.INT64_MAX]), "a": np.array([1]), "t": np.array([2])},
                       {"x": x}, ["y"])
    assert s.tolist() == [[2, 4]]


@pytest.mark.skipif(not os.path.isfile(REF_ONNX), reason="silero_vad_v6.onnx not available on this box")
def test_real_asset_graph_execution_on_the_reference_speech_fixture():
    nodes, inits, ins, outs = onnx_lite.load(REF_ONNX)
    doc, consts = _topology()
    assert nodes == doc["nodes"] and ins == doc["inputs"] and outs == doc["outputs"]      # the committed topology is the asset's
    assert {k: list(v.shape) for k, v in inits.items() if k not in consts} == doc["weight_shapes"]
    for k, v in consts.items():
        assert np.array_equal(inits[k], v)
    speech = np.load(os.path.join(GOLD, "speech_pcm.npz"))["pcm"].astype(np.float32)
    audio = np.concatenate([np.zeros(32000, np.float32), speech, np.zeros(31000, np.float32)])
    padded = np.pad(audio, (0, 512 - len(audio) % 512))
    win = silero.frame_windows(padded)
    z = np.zeros((1, 1, 128), np.float32)
    probs, hn, cn = onnx_exec.run(nodes, inits, {"input": win, "h": z, "c": z}, outs)
    ref, rh, rc = silero.forward(inits, win)
    gold = np.load(os.path.join(GOLD, "vad_speech_probs.npy"))
    print(f"real asset, {len(win)} windows: graph execution vs restatement {np.abs(probs - ref).max():.2e}, "
          f"vs the committed probabilities {np.abs(probs - gold).max():.2e}")
    assert np.abs(probs - ref).max() < 5e-6 and np.abs(hn.reshape(128) - rh).max() < 5e-6
    assert np.abs(probs - gold).max() < 1e-5
    assert probs[:55].max() < 0.05 and (probs[70:70 + len(speech) // 512 - 16] > 0.5).mean() > 0.7
