"""Minimal ONNX reader (no `onnx` / `onnxruntime` dependency): decodes the protobuf wire format of a ModelProto
far enough to list the graph's nodes (op type, inputs, outputs, int/float/ints/string attributes) and to
return its initializers as numpy arrays.  Used to load the Silero VAD network the reference ships as
`faster_whisper/assets/silero_vad_v6.onnx` (vad.py:288-292).

Field numbers (onnx.proto): ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5, .input = 11,
.output = 12; NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5;
AttributeProto.name = 1, .f = 2, .i = 3, .s = 4, .t = 5, .floats = 7, .ints = 8;
TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int64_data = 7, .name = 8, .raw_data = 9.
"""
import struct
from typing import Dict, List, Tuple

import numpy as np

_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 10: np.float16,
           11: np.float64}


def _varint(buf: bytes, i: int) -> Tuple[int, int]:
    value = shift = 0
    while True:
        b = buf[i]
        i += 1
        value |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return value, i


def _fields(buf: bytes) -> List[Tuple[int, int, object]]:
    """-> [(field number, wire type, value)]; length-delimited values stay bytes"""
    out, i = [], 0
    while i < len(buf):
        key, i = _varint(buf, i)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 2:
            n, i = _varint(buf, i)
            v, i = buf[i:i + n], i + n
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        out.append((field, wire, v))
    return out


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _ints(value, wire) -> List[int]:
    if wire == 0:
        return [_signed(value)]
    out, i = [], 0
    while i < len(value):      # packed repeated varints
        v, i = _varint(value, i)
        out.append(_signed(v))
    return out


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims, dtype, name, raw, floats, int64s = [], 1, "", None, [], []
    for f, w, v in _fields(buf):
        if f == 1:
            dims += _ints(v, w)
        elif f == 2:
            dtype = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 7:
            int64s += _ints(v, w)
    if dtype not in _DTYPES:
        raise ValueError(f"tensor '{name}': unsupported ONNX data type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, dtype=_DTYPES[dtype]).copy()
    elif floats:
        arr = np.asarray(floats, dtype=np.float32)
    else:
        arr = np.asarray(int64s, dtype=_DTYPES[dtype])
    return name, arr.reshape(dims) if dims else arr.reshape(())


def _attribute(buf: bytes):
    name, value = "", None
    for f, w, v in _fields(buf):
        if f == 1:
            name = v.decode()
        elif f == 2:
            value = struct.unpack("<f", v)[0]
        elif f == 3:
            value = _signed(v)
        elif f == 4:
            value = v.decode(errors="replace")
        elif f == 5:
            value = _tensor(v)[1]
        elif f == 7:
            value = (value or []) + (list(struct.unpack(f"<{len(v) // 4}f", v)) if w == 2
                                     else [struct.unpack("<f", v)[0]])
        elif f == 8:
            value = (value or []) + _ints(v, w)
    return name, value


def load(path: str) -> Tuple[List[dict], Dict[str, np.ndarray], List[str], List[str]]:
    """-> (nodes [{op, inputs, outputs, attrs}], initializers {name: array}, graph input names, output names)"""
    with open(path, "rb") as f:
        model = _fields(f.read())
    graphs = [v for f, w, v in model if f == 7 and w == 2]
    if not graphs:
        raise ValueError(f"{path}: not an ONNX model (no graph)")
    nodes, inits, inputs, outputs = [], {}, [], []
    for f, w, v in _fields(graphs[0]):
        if f == 1:
            node = dict(op="", inputs=[], outputs=[], attrs={})
            for nf, nw, nv in _fields(v):
                if nf == 1:
                    node["inputs"].append(nv.decode())
                elif nf == 2:
                    node["outputs"].append(nv.decode())
                elif nf == 4:
                    node["op"] = nv.decode()
                elif nf == 5:
                    k, val = _attribute(nv)
                    node["attrs"][k] = val
            nodes.append(node)
        elif f == 5:
            name, arr = _tensor(v)
            inits[name] = arr
        elif f in (11, 12):
            name = next((x.decode() for ff, ww, x in _fields(v) if ff == 1), "")
            (inputs if f == 11 else outputs).append(name)
    inputs = [n for n in inputs if n not in inits]
    return nodes, inits, inputs, outputs
