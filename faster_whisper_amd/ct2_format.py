"""Reader / writer for CTranslate2 converted Whisper model directories:
`model.bin` + `config.json`, the file set the reference downloads (utils.py:91-97).

[CT2-ext] The binary layout below is restated from the published CTranslate2 4.x converter
(`python/ctranslate2/specs/model_spec.py::_serialize`) — it cannot be checked against a real
file in this environment (no checkpoint on disk, no network), so every structural assumption is
validated while parsing and a wrong guess fails loudly instead of loading garbage:

    u32  binary_version (6)
    str  spec name ("WhisperSpec")          str = u16 length (incl. NUL) + bytes + NUL
    u32  spec revision
    u32  n_variables
    n x { str name; u8 rank; u32 dims[rank]; u8 dtype_id; u32 n_bytes; raw little-endian data }
    u32  n_aliases
    n x { str alias; str target }
dtype ids: 0 float32, 1 int8, 2 int16, 3 int32, 4 float16, 5 bfloat16.
int8 weights carry a float32 `<name>_scale` per output row (value = int8 / scale).
"""
import json
import os
import struct
from typing import Dict, Tuple

import numpy as np

from .config import WhisperConfig, get_config

_DTYPES = {0: np.float32, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.float16}
_DTYPE_IDS = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3,
              np.dtype(np.float16): 4}


def _read_str(f) -> str:
    (n,) = struct.unpack("<H", f.read(2))
    raw = f.read(n)
    if len(raw) != n or n == 0 or raw[-1] != 0:
        raise ValueError("model.bin: malformed string field")
    return raw[:-1].decode("utf-8")


def _write_str(f, s: str):
    b = s.encode("utf-8")
    f.write(struct.pack("<H", len(b) + 1))
    f.write(b)
    f.write(b"\x00")


def read_model_bin(path: str) -> Tuple[str, int, Dict[str, np.ndarray], Dict[str, str]]:
    """-> (spec name, revision, variables, aliases)"""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        (version,) = struct.unpack("<I", f.read(4))
        if not 1 <= version <= 16:
            raise ValueError(f"model.bin: implausible binary version {version}")
        spec = _read_str(f)
        (revision,) = struct.unpack("<I", f.read(4))
        (n_vars,) = struct.unpack("<I", f.read(4))
        if n_vars > 100000:
            raise ValueError("model.bin: implausible variable count")
        variables = {}
        for _ in range(n_vars):
            name = _read_str(f)
            (rank,) = struct.unpack("<B", f.read(1))
            dims = struct.unpack(f"<{rank}I", f.read(4 * rank)) if rank else ()
            (dt,) = struct.unpack("<B", f.read(1))
            (nbytes,) = struct.unpack("<I", f.read(4))
            if dt == 5:
                raw = np.frombuffer(f.read(nbytes), dtype=np.uint16)
                arr = (raw.astype(np.uint32) << 16).view(np.float32)       # bfloat16 -> float32
            elif dt in _DTYPES:
                arr = np.frombuffer(f.read(nbytes), dtype=_DTYPES[dt])
            else:
                raise ValueError(f"model.bin: unknown dtype id {dt} for '{name}'")
            count = int(np.prod(dims)) if rank else 1
            if arr.size != count or f.tell() > size:
                raise ValueError(f"model.bin: variable '{name}' has {arr.size} elements, shape says {count}")
            variables[name] = arr.reshape(dims) if rank else arr.reshape(())
        aliases = {}
        tail = f.read(4)
        if len(tail) == 4:
            (n_alias,) = struct.unpack("<I", tail)
            for _ in range(n_alias):
                a = _read_str(f)
                aliases[a] = _read_str(f)
    return spec, revision, variables, aliases


def write_model_bin(path: str, variables: Dict[str, np.ndarray], aliases: Dict[str, str] = None,
                    spec: str = "WhisperSpec", revision: int = 3, version: int = 6):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", version))
        _write_str(f, spec)
        f.write(struct.pack("<I", revision))
        f.write(struct.pack("<I", len(variables)))
        for name, a in variables.items():
            a = np.ascontiguousarray(a)
            _write_str(f, name)
            f.write(struct.pack("<B", a.ndim))
            for d in a.shape:
                f.write(struct.pack("<I", d))
            f.write(struct.pack("<B", _DTYPE_IDS[a.dtype]))
            f.write(struct.pack("<I", a.nbytes))
            f.write(a.tobytes())
        aliases = aliases or {}
        f.write(struct.pack("<I", len(aliases)))
        for k, v in aliases.items():
            _write_str(f, k)
            _write_str(f, v)


# ---- name mapping: CTranslate2 WhisperSpec <-> this engine (faster_whisper_amd/weights.py) ----
def _layer_map(side: str, i: int):
    p, q = f"{side}/layer_{i}/", f"{'enc' if side == 'encoder' else 'dec'}.{i}."
    m = {p + "self_attention/layer_norm/gamma": q + "ln1.g", p + "self_attention/layer_norm/beta": q + "ln1.b",
         p + "self_attention/linear_0/weight": q + ("attn.qkv.w" if side == "encoder" else "self.qkv.w"),
         p + "self_attention/linear_0/bias": q + ("attn.qkv.b" if side == "encoder" else "self.qkv.b"),
         p + "self_attention/linear_1/weight": q + ("attn.out.w" if side == "encoder" else "self.out.w"),
         p + "self_attention/linear_1/bias": q + ("attn.out.b" if side == "encoder" else "self.out.b"),
         p + "ffn/linear_0/weight": q + "ffn1.w", p + "ffn/linear_0/bias": q + "ffn1.b",
         p + "ffn/linear_1/weight": q + "ffn2.w", p + "ffn/linear_1/bias": q + "ffn2.b"}
    if side == "encoder":
        m.update({p + "ffn/layer_norm/gamma": q + "ln2.g", p + "ffn/layer_norm/beta": q + "ln2.b"})
    else:
        m.update({p + "attention/layer_norm/gamma": q + "ln2.g", p + "attention/layer_norm/beta": q + "ln2.b",
                  p + "attention/linear_0/weight": q + "cross.q.w", p + "attention/linear_0/bias": q + "cross.q.b",
                  p + "attention/linear_1/weight": q + "cross.kv.w", p + "attention/linear_1/bias": q + "cross.kv.b",
                  p + "attention/linear_2/weight": q + "cross.out.w", p + "attention/linear_2/bias": q + "cross.out.b",
                  p + "ffn/layer_norm/gamma": q + "ln3.g", p + "ffn/layer_norm/beta": q + "ln3.b"})
    return m


def name_map(n_enc: int, n_dec: int) -> Dict[str, str]:
    m = {"encoder/conv1/weight": "enc.conv1.w", "encoder/conv1/bias": "enc.conv1.b",
         "encoder/conv2/weight": "enc.conv2.w", "encoder/conv2/bias": "enc.conv2.b",
         "encoder/position_encodings/encodings": "enc.pos",
         "encoder/layer_norm/gamma": "enc.ln_post.g", "encoder/layer_norm/beta": "enc.ln_post.b",
         "decoder/embeddings/weight": "dec.tok_emb", "decoder/position_encodings/encodings": "dec.pos",
         "decoder/layer_norm/gamma": "dec.ln.g", "decoder/layer_norm/beta": "dec.ln.b"}
    for i in range(n_enc):
        m.update(_layer_map("encoder", i))
    for i in range(n_dec):
        m.update(_layer_map("decoder", i))
    return m


def _dequant(variables: Dict[str, np.ndarray], name: str) -> np.ndarray:
    a = variables[name]
    if a.dtype == np.int8:
        scale = variables.get(name + "_scale")
        if scale is None:
            raise ValueError(f"model.bin: int8 tensor '{name}' has no '{name}_scale'")
        return a.astype(np.float32) / scale.astype(np.float32).reshape(-1, *([1] * (a.ndim - 1)))
    if a.dtype == np.int16:
        scale = variables.get(name + "_scale")
        return a.astype(np.float32) / float(scale)
    return a


def load_ct2_model_dir(path: str) -> Tuple[WhisperConfig, Dict[str, np.ndarray]]:
    """CTranslate2 Whisper directory -> (WhisperConfig, engine weight dict)."""
    spec, _rev, variables, aliases = read_model_bin(os.path.join(path, "model.bin"))
    if "Whisper" not in spec:
        raise ValueError(f"model.bin holds a '{spec}', not a Whisper model")
    for alias, target in aliases.items():
        if target in variables and alias not in variables:
            variables[alias] = variables[target]
    import re

    def n_layers(side):
        idx = [int(mm.group(1)) for mm in (re.match(rf"^{side}/layer_(\d+)/", k) for k in variables) if mm]
        if not idx:
            raise ValueError(f"model.bin has no '{side}/layer_N/...' variables")
        return 1 + max(idx)
    n_enc, n_dec = n_layers("encoder"), n_layers("decoder")
    conv1 = variables["encoder/conv1/weight"]
    d, n_mels = int(conv1.shape[0]), int(conv1.shape[1])
    n_vocab = int(variables["decoder/embeddings/weight"].shape[0])
    base = {51864: "tiny.en", 51865: "tiny", 51866: "large-v3"}.get(n_vocab)
    if base is None:
        raise ValueError(f"unsupported Whisper vocabulary size {n_vocab}")
    proto = get_config(base)
    cfg = WhisperConfig(name=os.path.basename(os.path.normpath(path)), n_mels=n_mels, d_model=d, n_heads=d // 64,
                        n_enc_layers=n_enc, n_dec_layers=n_dec, n_vocab=n_vocab, is_multilingual=proto.is_multilingual,
                        n_text_ctx=int(variables["decoder/position_encodings/encodings"].shape[0]),
                        eot=proto.eot, sot=proto.sot, lang_begin=proto.lang_begin, n_langs=proto.n_langs,
                        translate=proto.translate, transcribe=proto.transcribe, sot_lm=proto.sot_lm,
                        sot_prev=proto.sot_prev, no_speech=proto.no_speech, no_timestamps=proto.no_timestamps,
                        timestamp_begin=proto.timestamp_begin, suppress_begin=proto.suppress_begin)
    cj = os.path.join(path, "config.json")
    if os.path.isfile(cj):
        with open(cj) as f:
            j = json.load(f)
        if j.get("suppress_ids_begin"):
            cfg.suppress_begin = tuple(int(t) for t in j["suppress_ids_begin"][:8])
        if j.get("alignment_heads"):
            cfg.alignment_heads = [(int(l), int(h)) for l, h in j["alignment_heads"]]
        if j.get("lang_ids"):
            cfg.lang_begin, cfg.n_langs = int(min(j["lang_ids"])), len(j["lang_ids"])
    weights = {}
    for ct2_name, ours in name_map(n_enc, n_dec).items():
        if ct2_name not in variables:
            raise ValueError(f"model.bin lacks '{ct2_name}'")
        weights[ours] = np.ascontiguousarray(_dequant(variables, ct2_name))
    return cfg, weights


def save_ct2_model_dir(path: str, cfg: WhisperConfig, weights: Dict[str, np.ndarray]):
    """inverse of load_ct2_model_dir (float tensors only) — used by the round-trip test"""
    os.makedirs(path, exist_ok=True)
    inv = {v: k for k, v in name_map(cfg.n_enc_layers, cfg.n_dec_layers).items()}
    variables = {inv[k]: np.ascontiguousarray(v) for k, v in weights.items()}
    write_model_bin(os.path.join(path, "model.bin"), variables,
                    aliases={"decoder/projection/weight": "decoder/embeddings/weight"})
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"suppress_ids_begin": list(cfg.suppress_begin),
                   "alignment_heads": [list(x) for x in cfg.alignment_heads],
                   "lang_ids": list(range(cfg.lang_begin, cfg.lang_begin + cfg.n_langs))}, f)
