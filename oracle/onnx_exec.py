"""TEST INFRASTRUCTURE — a GENERIC executor for small ONNX graphs: walks the node list of a model file (read with the
product's protobuf reader, faster_whisper_amd/onnx_lite.py) and evaluates every node by its operator's definition in
the ONNX operator specification, on torch's / numpy's own kernels.  It knows nothing about Silero: no layer names, no
shapes, no gate bookkeeping beyond what the `LSTM` operator's specification says.

Why it exists: `oracle/silero.py` is a HAND restatement of the graph of the reference's VAD asset
(`faster_whisper/assets/silero_vad_v6.onnx`, run by onnxruntime in vad.py:288-351) and onnxruntime is not installed,
so the restatement had nothing to be checked against.  Executing the reference's own model file node by node — its
topology, attributes and constants as stored, torch's conv1d / LSTM kernels instead of hand-written loops — is an
independent derivation of the same outputs (tests/test_oracle_silero_graph.py compares the two on the reference's
speech fixture and on noise, and pins a committed golden vector made by tests/golden/make_silero_graph_golden.py).
It is NOT onnxruntime: the pin is "the asset's graph under the operator specification", stated as such in DESIGN.md.

Operators (the 13 the asset uses; anything else raises): Pad (reflect / constant / edge), Unsqueeze, Squeeze, Reshape,
Transpose, Slice, Conv (1-D), Pow, Add, Sqrt, Relu, Sigmoid, LSTM (forward, one direction, default activations).
Only tests/ import this module.
"""
from typing import Dict, List, Sequence

import numpy as np
import torch

INT64_MAX = (1 << 63) - 1


def _axes(v, rank):
    return sorted(int(a) % rank for a in np.asarray(v).reshape(-1))


def _pad(x, pads, mode, value=0.0):
    r = x.ndim
    p = [int(v) for v in np.asarray(pads).reshape(-1)]
    if len(p) != 2 * r:
        raise ValueError(f"Pad: {len(p)} pad values for a rank-{r} input")
    width = [(p[i], p[i + r]) for i in range(r)]              # ONNX: all begins, then all ends
    if mode == "reflect":
        return np.pad(x, width, mode="reflect")               # no edge duplication — the ONNX definition
    if mode == "edge":
        return np.pad(x, width, mode="edge")
    return np.pad(x, width, mode="constant", constant_values=value)


def _slice(x, starts, ends, axes=None, steps=None):
    r = x.ndim
    starts = [int(v) for v in np.asarray(starts).reshape(-1)]
    ends = [int(v) for v in np.asarray(ends).reshape(-1)]
    axes = list(range(len(starts))) if axes is None else [int(a) % r for a in np.asarray(axes).reshape(-1)]
    steps = [1] * len(starts) if steps is None else [int(s) for s in np.asarray(steps).reshape(-1)]
    idx = [slice(None)] * r
    for s, e, a, st in zip(starts, ends, axes, steps):
        if st <= 0:
            raise ValueError("Slice: only positive steps")
        n = x.shape[a]
        s = min(max(s + n if s < 0 else s, 0), n)             # clamp as the specification says (INT64_MAX = "to the end")
        e = min(max(e + n if e < 0 else e, 0), n)
        idx[a] = slice(s, e, st)
    return x[tuple(idx)]


def _conv(x, w, b, attrs):
    if w.ndim != 3:
        raise ValueError("Conv: only 1-D convolutions")
    pads = list(attrs.get("pads", [0, 0]))
    if pads[0] != pads[1]:
        raise ValueError("Conv: asymmetric padding")
    if attrs.get("auto_pad", "NOTSET") not in ("NOTSET", b"NOTSET"):
        raise ValueError("Conv: auto_pad")
    y = torch.nn.functional.conv1d(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(w)),
                                   None if b is None else torch.from_numpy(np.ascontiguousarray(b)),
                                   stride=int(attrs.get("strides", [1])[0]), padding=int(pads[0]),
                                   dilation=int(attrs.get("dilations", [1])[0]), groups=int(attrs.get("group", 1)))
    return y.numpy()


def _lstm(x, w, r, b, seq_lens, h0, c0, attrs):
    """ONNX LSTM, forward direction: X [seq, batch, in], W [1, 4H, in] and R [1, 4H, H] with gate order i o f c,
    B [1, 8H] = Wb | Rb, initial_h / initial_c [1, batch, H]; default activations (sigmoid, tanh, tanh), no peepholes,
    no clip.  -> Y [seq, 1, batch, H], Y_h [1, batch, H], Y_c [1, batch, H].  Evaluated by torch.nn.LSTM, whose gate
    order is i f g o: the rows are re-ordered, nothing else."""
    if attrs.get("direction", "forward") not in ("forward", b"forward") or int(attrs.get("layout", 0)) != 0:
        raise ValueError("LSTM: forward direction, layout 0 only")
    if seq_lens is not None or "clip" in attrs or "activations" in attrs or int(attrs.get("input_forget", 0)):
        raise ValueError("LSTM: sequence_lens / clip / activations / input_forget are not supported")
    H = int(attrs["hidden_size"])
    if w.shape[0] != 1 or w.shape[1] != 4 * H:
        raise ValueError("LSTM: W must be [1, 4 * hidden_size, input_size]")
    order = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(3 * H, 4 * H), np.arange(H, 2 * H)])  # i f c o
    net = torch.nn.LSTM(input_size=w.shape[2], hidden_size=H, num_layers=1, bias=True, batch_first=False)
    bias = np.zeros(8 * H, np.float32) if b is None else np.asarray(b, np.float32).reshape(8 * H)
    with torch.no_grad():
        net.weight_ih_l0.copy_(torch.from_numpy(np.ascontiguousarray(w[0][order])))
        net.weight_hh_l0.copy_(torch.from_numpy(np.ascontiguousarray(r[0][order])))
        net.bias_ih_l0.copy_(torch.from_numpy(np.ascontiguousarray(bias[:4 * H][order])))
        net.bias_hh_l0.copy_(torch.from_numpy(np.ascontiguousarray(bias[4 * H:][order])))
        batch = x.shape[1]
        h = torch.zeros(1, batch, H) if h0 is None else torch.from_numpy(np.ascontiguousarray(h0, dtype=np.float32))
        c = torch.zeros(1, batch, H) if c0 is None else torch.from_numpy(np.ascontiguousarray(c0, dtype=np.float32))
        y, (hn, cn) = net(torch.from_numpy(np.ascontiguousarray(x)), (h, c))
    return y.numpy()[:, None, :, :], hn.numpy(), cn.numpy()


def run(nodes: List[dict], inits: Dict[str, np.ndarray], feeds: Dict[str, np.ndarray], outputs: Sequence[str]):
    """nodes / inits as returned by onnx_lite.load (file order = a valid topological order, as ONNX requires)"""
    env = dict(inits)
    env.update({k: np.asarray(v) for k, v in feeds.items()})

    def get(name):
        if name == "":
            return None
        if name not in env:
            raise KeyError(f"tensor '{name}' is read before it is produced")
        return env[name]

    for nd in nodes:
        op, a = nd["op"], nd["attrs"]
        i = [get(n) for n in nd["inputs"]]
        if op == "Pad":
            mode = a.get("mode", "constant")
            out = [_pad(i[0], i[1], mode if isinstance(mode, str) else mode.decode(),
                        float(i[2]) if len(i) > 2 and i[2] is not None else 0.0)]
        elif op == "Unsqueeze":
            y = i[0]
            for ax in _axes(i[1], i[0].ndim + np.asarray(i[1]).size):
                y = np.expand_dims(y, ax)
            out = [y]
        elif op == "Squeeze":
            out = [np.squeeze(i[0], axis=tuple(_axes(i[1], i[0].ndim))) if len(i) > 1 and i[1] is not None else np.squeeze(i[0])]
        elif op == "Reshape":
            shape = [int(v) for v in np.asarray(i[1]).reshape(-1)]
            if not int(a.get("allowzero", 0)):
                shape = [i[0].shape[k] if v == 0 else v for k, v in enumerate(shape)]
            out = [i[0].reshape(shape)]
        elif op == "Transpose":
            out = [np.transpose(i[0], a.get("perm", list(range(i[0].ndim))[::-1]))]
        elif op == "Slice":
            out = [_slice(*i)]
        elif op == "Conv":
            out = [_conv(i[0], i[1], i[2] if len(i) > 2 else None, a)]
        elif op == "Pow":
            out = [np.power(i[0], np.asarray(i[1], dtype=i[0].dtype))]
        elif op == "Add":
            out = [i[0] + i[1]]
        elif op == "Sqrt":
            out = [np.sqrt(i[0])]
        elif op == "Relu":
            out = [np.maximum(i[0], 0)]
        elif op == "Sigmoid":
            out = [torch.sigmoid(torch.from_numpy(np.ascontiguousarray(i[0]))).numpy()]
        elif op == "LSTM":
            i += [None] * (8 - len(i))
            if i[7] is not None:
                raise ValueError("LSTM: peepholes are not supported")
            out = list(_lstm(i[0], i[1], i[2], i[3], i[4], i[5], i[6], a))
        else:
            raise NotImplementedError(f"ONNX operator {op}")
        for name, val in zip(nd["outputs"], out):
            if name:
                env[name] = val
    return [env[n] for n in outputs]
