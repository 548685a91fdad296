"""TEST INFRASTRUCTURE — numpy restatement of the Silero VAD v6 network (the ONNX asset the reference runs with
onnxruntime, faster_whisper/vad.py:288-351) and of `SileroVADModel.__call__`'s window framing.

PARITY vs onnxruntime UNPINNED: `onnxruntime` is not installed, so no output of the reference's own VAD run can be
produced here.  What the restatement IS pinned to (round 6): the asset's own graph executed generically —
`oracle/onnx_exec.py` walks the 25 nodes of `silero_vad_v6.onnx` and evaluates each by its ONNX operator definition on
torch's conv1d / LSTM kernels, knowing nothing of this file — `tests/test_oracle_silero_graph.py`: 5.8e-7 on the
reference's speech fixture with the real weights, <= 1.6e-6 (probabilities, h, c) with random weights on the committed
topology.  The graph semantics below follow the ONNX operator specification (Pad reflect, Conv, Slice, LSTM with gate
order i, o, f, c) applied to the node list of `silero_vad_v6.onnx`:

    input [N, 576] (64 context + 512 new samples)
    Pad reflect 128 | 128                     -> [N, 832]
    Conv basis [258, 1, 256], stride 128      -> [N, 258, 5]; Slice drops frame 0 -> real [:, :129, 1:], imag [:, 129:, 1:]
    sqrt(re^2 + im^2)                         -> [N, 129, 4]
    Conv 129->128 k3 p1 s1, ReLU              -> [N, 128, 4]
    Conv 128->64  k3 p1 s2, ReLU              -> [N, 64, 2]
    Conv 64->64   k3 p1 s2, ReLU              -> [N, 64, 1]
    Conv 64->128  k3 p1 s1, ReLU              -> [N, 128, 1]
    Transpose -> LSTM over the N windows as the SEQUENCE (batch 1, hidden 128, state h / c carried)
    ReLU, Conv 128->1 k1, Sigmoid             -> speech_probs [N]

Plausibility anchor (tests/test_vad_network.py, build container only): on the reference's own speech fixture the
probabilities are high on speech and low on digital silence.  Only tests/ import this module.
"""
from typing import Dict, Optional, Tuple

import numpy as np

NAMES = dict(
    basis="encoder.feature_extractor.forward_basis_buffer",
    conv_w=[f"encoder.conv_layers.{i}.weight" for i in range(4)],
    conv_b=[f"encoder.conv_layers.{i}.bias" for i in range(4)],
    dec_w="decoder.conv1d.weight", dec_b="decoder.conv1d.bias")
STRIDES = (1, 2, 2, 1)


def frame_windows(audio: np.ndarray, num_samples: int = 512, context: int = 64) -> np.ndarray:
    """SileroVADModel.__call__ framing (vad.py:318-336): [n, 576] = 64 samples of context (the tail of the
    previous window, zeros for the first) + the 512-sample window.  Reference quirk kept: the context slice is
    a VIEW, so `context[-1] = 0` also zeroes the last 64 samples of the last window of the (padded) audio."""
    assert audio.ndim == 1 and audio.shape[0] % num_samples == 0
    win = np.array(audio, dtype=np.float32).reshape(-1, num_samples)
    win[-1, -context:] = 0
    ctx = np.roll(win[:, -context:], 1, axis=0)
    return np.concatenate([ctx, win], axis=1)


def lstm_weights(inits: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """the three anonymous LSTM initializers by shape: W [1,512,128] (first), R [1,512,128] (second), B [1,1024]"""
    mats = [v for k, v in inits.items() if v.shape == (1, 512, 128)]
    bias = [v for k, v in inits.items() if v.shape == (1, 1024)]
    assert len(mats) == 2 and len(bias) == 1
    return mats[0][0], mats[1][0], bias[0][0]


def _conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """x [N, Cin, T], w [Cout, Cin, 3], zero padding 1"""
    n, cin, t = x.shape
    xp = np.zeros((n, cin, t + 2), dtype=np.float32)
    xp[:, :, 1:-1] = x
    t_out = (t + 2 - 3) // stride + 1
    out = np.empty((n, w.shape[0], t_out), dtype=np.float32)
    for j in range(t_out):
        patch = xp[:, :, j * stride:j * stride + 3].reshape(n, -1)          # [N, Cin*3]
        out[:, :, j] = patch @ w.reshape(w.shape[0], -1).T + b
    return out


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def forward(inits: Dict[str, np.ndarray], windows: np.ndarray, h: Optional[np.ndarray] = None,
            c: Optional[np.ndarray] = None):
    """windows [N, 576] float32 -> (speech_probs [N], h [128], c [128])"""
    x = np.asarray(windows, dtype=np.float32)
    n = x.shape[0]
    xp = np.pad(x, ((0, 0), (128, 128)), mode="reflect")
    basis = inits[NAMES["basis"]][:, 0, :]                                   # [258, 256]
    frames = np.stack([xp[:, 128 * t:128 * t + 256] for t in range(1, 5)], axis=1)   # [N, 4, 256]
    spec = frames @ basis.T                                                  # [N, 4, 258]
    mag = np.sqrt(spec[..., :129] ** 2 + spec[..., 129:] ** 2).transpose(0, 2, 1)   # [N, 129, 4]
    y = mag.astype(np.float32)
    for i in range(4):
        y = np.maximum(_conv1d(y, inits[NAMES["conv_w"][i]], inits[NAMES["conv_b"][i]], STRIDES[i]), 0)
    feat = y[:, :, 0]                                                        # [N, 128]
    W, R, B = lstm_weights(inits)
    gx = feat @ W.T + B[:512] + B[512:]                                      # [N, 512]
    h = np.zeros(128, np.float32) if h is None else np.asarray(h, np.float32).reshape(128).copy()
    c = np.zeros(128, np.float32) if c is None else np.asarray(c, np.float32).reshape(128).copy()
    dec_w = inits[NAMES["dec_w"]].reshape(128)
    dec_b = np.float32(inits[NAMES["dec_b"]].reshape(()))
    probs = np.empty(n, dtype=np.float32)
    for t in range(n):
        g = gx[t] + R @ h
        i_g, o_g, f_g = _sigmoid(g[:128]), _sigmoid(g[128:256]), _sigmoid(g[256:384])
        c = f_g * c + i_g * np.tanh(g[384:])
        h = (o_g * np.tanh(c)).astype(np.float32)
        probs[t] = _sigmoid(np.float32(np.maximum(h, 0) @ dec_w + dec_b))
    return probs, h, c
