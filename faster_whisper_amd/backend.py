"""`ctranslate2.models.Whisper`-compatible front of the MI355X engine.

This is the host-side mirror of the reference's backend interface, exactly as it is used
at the call sites in faster_whisper/transcribe.py:
    constructor :689-698, .is_multilingual/.n_mels/.device/.device_index :379,484,1394,
    .encode :1400, .generate :222-236 / :1446-1459, .detect_language :215,1193,1823,
    .align :1709-1715, StorageView.from_array :1875,
    WhisperGenerationResult(.sequences_ids,.scores,.no_speech_prob) :241-248,
    WhisperAlignmentResult(.alignments,.text_token_probs) :1718-1719.
Same names, argument meaning and error behaviour (ValueError for bad arguments,
RuntimeError otherwise).  All arithmetic happens in libfwamd.so (HIP, gfx950); this file
only marshals.
"""
import ctypes as C
import itertools
import json
import os
import queue
import threading
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .config import WhisperConfig, get_config
from .weights import synthetic_weights, weight_shapes

_COMPUTE_TYPES = {
    "default": _lib.COMPUTE_FLOAT16, "auto": _lib.COMPUTE_FLOAT16, "float16": _lib.COMPUTE_FLOAT16,
    "int8_float16": _lib.COMPUTE_INT8_FLOAT16, "int8": _lib.COMPUTE_INT8_FLOAT16,
}


class StorageView:
    """N-d tensor handle: either a borrowed host numpy array (from_array) or a
    device-resident encoder output owned by the engine."""

    def __init__(self, array: Optional[np.ndarray] = None, handle=None, owner=None, shape=None, parts=None):
        self._array = array
        self._handle = handle
        self._owner = owner  # the _Replica that produced the handle
        # an encoder output of more chunks than the engine's max_batch is a list of device tensors (CTranslate2
        # takes any batch size; the engine's workspaces are sized once)
        self._parts = parts
        if parts:
            shape = (sum(p._shape[0] for p in parts),) + tuple(parts[0]._shape[1:])
        self._shape = tuple(shape) if shape is not None else (tuple(array.shape) if array is not None else ())

    @classmethod
    def from_array(cls, array) -> "StorageView":
        array = np.asarray(array)
        if not array.flags["C_CONTIGUOUS"]:
            raise ValueError("StorageView.from_array requires a C-contiguous array")
        return cls(array=array)

    @property
    def shape(self):
        return list(self._shape)

    @property
    def device(self) -> str:
        return "cpu" if self._handle is None and not self._parts else "cuda"

    @property
    def dtype(self):
        return np.float32 if self._handle is None else np.float16

    def to_numpy(self) -> np.ndarray:
        """float32 copy on the host (for device tensors: the to_cpu path)."""
        if self._parts:
            return np.concatenate([p.to_numpy() for p in self._parts], axis=0)
        if self._handle is None:
            return np.asarray(self._array)
        out = np.empty(self._shape, dtype=np.float32)
        _lib.check(_lib.load().fw_tensor_to_host(self._owner.handle, self._handle, _lib.ptr(out)))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.to_numpy()
        return a.astype(dtype) if dtype is not None else a

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.load().fw_tensor_free(self._handle)
            except Exception:
                pass
            self._handle = None


class WhisperGenerationResult:
    def __init__(self, sequences_ids, scores, no_speech_prob):
        self.sequences_ids = sequences_ids
        self.sequences = [[str(t) for t in s] for s in sequences_ids]
        self.scores = scores
        self.no_speech_prob = no_speech_prob

    def __repr__(self):
        return (f"WhisperGenerationResult(sequences_ids={self.sequences_ids}, scores={self.scores}, "
                f"no_speech_prob={self.no_speech_prob})")


class WhisperAlignmentResult:
    def __init__(self, alignments, text_token_probs):
        self.alignments = alignments
        self.text_token_probs = text_token_probs


class _Replica:
    """one fw_model on one GPU"""

    def __init__(self, cfg: WhisperConfig, weights: Optional[Dict[str, np.ndarray]], compute_type: int,
                 device_index: int, max_batch: int, max_beam: int, blob_dev: Optional[Tuple[int, int]] = None):
        lib = _lib.load()
        self.cfg = cfg
        self.device_index = device_index
        self.handle = C.c_void_p()
        ccfg = cfg.to_c()
        if blob_dev is not None:
            ptr, nbytes = blob_dev
            _lib.check(lib.fw_model_create_from_blob_dev(C.byref(ccfg), C.c_void_p(ptr), nbytes, compute_type,
                                                         device_index, max_batch, max_beam, C.byref(self.handle)))
            return
        arr, keep = make_weight_array(cfg, weights)
        _lib.check(lib.fw_model_create(C.byref(ccfg), arr, len(keep), compute_type, device_index, max_batch,
                                       max_beam, C.byref(self.handle)))

    def close(self):
        if self.handle:
            _lib.load().fw_model_free(self.handle)
            self.handle = C.c_void_p()


def make_weight_array(cfg: WhisperConfig, weights: Dict[str, np.ndarray]):
    shapes = weight_shapes(cfg)
    missing = [n for n in shapes if n not in weights]
    if missing:
        raise ValueError(f"missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    arr = (_lib.FwWeight * len(shapes))()
    keep = []
    for i, (name, shape) in enumerate(shapes.items()):
        a = weights[name]
        if a.dtype not in (np.float16, np.float32):
            a = a.astype(np.float32)
        a = np.ascontiguousarray(a)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"weight {name}: shape {a.shape}, expected {shape}")
        keep.append(a)
        arr[i].name = name.encode()
        arr[i].data = a.ctypes.data
        arr[i].dtype = _lib.FW_DT_F16 if a.dtype == np.float16 else _lib.FW_DT_F32
        arr[i].ndim = a.ndim
        for k in range(a.ndim):
            arr[i].dims[k] = a.shape[k]
    return arr, keep


def pack_blob(cfg: WhisperConfig, weights: Dict[str, np.ndarray], compute_type: int = 0) -> np.ndarray:
    """Host image of the device weight blob (what rank 0 broadcasts over RCCL at load)."""
    lib = _lib.load()
    arr, keep = make_weight_array(cfg, weights)
    size = C.c_int64()
    h = C.c_void_p()
    ccfg = cfg.to_c()
    _lib.check(lib.fw_pack_blob_size(C.byref(ccfg), arr, len(keep), compute_type, C.byref(size), C.byref(h)))
    out = np.empty(size.value, dtype=np.uint8)
    try:
        _lib.check(lib.fw_pack_blob_copy(h, _lib.ptr(out), size.value))
    finally:
        lib.fw_pack_blob_free(h)
    return out


def load_model_dir(path: str) -> Tuple[WhisperConfig, Dict[str, np.ndarray]]:
    """fwamd model directory: fwamd_config.json + weights.safetensors (see save_model_dir)."""
    cfg_path = os.path.join(path, "fwamd_config.json")
    if not os.path.isfile(cfg_path):
        if os.path.isfile(os.path.join(path, "model.bin")):
            # a CTranslate2 converted directory (what the reference downloads, utils.py:91-97)
            from .ct2_format import load_ct2_model_dir
            return load_ct2_model_dir(path)
        raise RuntimeError(f"{path}: neither fwamd_config.json nor model.bin found")
    with open(cfg_path) as f:
        j = json.load(f)
    j["suppress_begin"] = tuple(j.get("suppress_begin", ()))
    j["alignment_heads"] = [tuple(x) for x in j.get("alignment_heads", [])]
    cfg = WhisperConfig(**j)
    from safetensors.numpy import load_file
    weights = load_file(os.path.join(path, "weights.safetensors"))
    return cfg, weights


def save_model_dir(path: str, cfg: WhisperConfig, weights: Dict[str, np.ndarray]):
    from dataclasses import asdict
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "fwamd_config.json"), "w") as f:
        json.dump(asdict(cfg), f)
    save_file({k: np.ascontiguousarray(v) for k, v in weights.items()}, os.path.join(path, "weights.safetensors"))


def call_seed(n: int) -> int:
    """seed of the n-th sampling call of a model (splitmix64 of the call counter)"""
    z = (n * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class Whisper:
    """Drop-in for ctranslate2.models.Whisper on MI355X.

    model_path:  a fwamd model directory, or "synthetic:<geometry>[:seed=<n>]" for seeded
                 random weights of a known geometry (benchmark / tests).
    files:       optional in-memory model: {"config": WhisperConfig, "weights": {name: ndarray}}
                 (the reference forwards its `files=` argument here, transcribe.py:696).
    """

    def __init__(self, model_path: str, device: str = "auto", device_index: Union[int, List[int]] = 0,
                 compute_type: str = "default", intra_threads: int = 0, inter_threads: int = 1,
                 files: Optional[dict] = None, max_batch_size: int = 16, max_beam_size: int = 5,
                 blob_dev: Optional[Tuple[int, int]] = None, **kwargs):
        if device not in ("auto", "cuda", "gpu", "hip"):
            raise ValueError(
                f"unsupported device '{device}': this backend runs on AMD GPUs only (device='cuda'/'auto'); "
                "there is no CPU path")
        if compute_type not in _COMPUTE_TYPES:
            raise ValueError(f"unsupported compute_type '{compute_type}' (supported: float16, int8_float16)")
        # the RESOLVED type, as ctranslate2's `compute_type` property reports it ("int8" on a device with fp16 -> int8_float16)
        self._compute_type_name = {"default": "float16", "auto": "float16",
                                   "int8": "int8_float16"}.get(compute_type, compute_type)
        ct = _COMPUTE_TYPES[compute_type]
        if files and "config" in files:
            cfg, weights = files["config"], files.get("weights")
        elif isinstance(model_path, str) and model_path.startswith("synthetic:"):
            parts = model_path.split(":")
            cfg = get_config(parts[1])
            seed = 1234
            for p in parts[2:]:
                if p.startswith("seed="):
                    seed = int(p[5:])
            weights = None if blob_dev is not None else synthetic_weights(cfg, seed)
        else:
            cfg, weights = load_model_dir(model_path)
        self._cfg = cfg
        self._max_batch = int(max_batch_size)
        idx = [device_index] if isinstance(device_index, int) else list(device_index)
        self._device_index = idx
        self._lib = _lib.load()
        # one replica per (device, worker): the `inter_threads` workers of a device share ONE copy of the
        # weights in HBM and own a stream + workspaces each (CTranslate2 replica pool semantics,
        # transcribe.py:645-657), so concurrent transcribe() threads overlap on the GPU
        # The workers of a device form a DECODE GROUP (include/fwamd.h): they share one decode workspace sized
        # for all of their batches, and generate() calls that arrive concurrently are merged into one decode run
        # whose rows share every weight byte streamed per step.  decode_group=False keeps a decoder per worker.
        self._replicas = []
        decode_group = bool(kwargs.pop("decode_group", True))
        # merge_wait_ms: how long the leader of a decode run waits for workers that are still encoding
        # (fwamd.h: fw_model_set_merge_wait; None = one encoder pass, 0 = never: lowest latency per call)
        merge_wait_ms = kwargs.pop("merge_wait_ms", None)
        merge_fill = kwargs.pop("merge_fill_percent", None)
        for i in idx:
            primary = _Replica(cfg, weights, ct, i, max_batch_size, max_beam_size, blob_dev)
            self._replicas.append(primary)
            if inter_threads > 1:
                p, n = C.c_void_p(), C.c_int64()
                _lib.check(self._lib.fw_model_blob(primary.handle, C.byref(p), C.byref(n)))
                if decode_group:
                    _lib.check(self._lib.fw_model_set_decode_batch(primary.handle, inter_threads * max_batch_size))
                for _ in range(inter_threads - 1):
                    r = _Replica(cfg, None, ct, i, max_batch_size, max_beam_size, (p.value, n.value))
                    if decode_group:
                        _lib.check(self._lib.fw_model_join_decoder(r.handle, primary.handle))
                    self._replicas.append(r)
            if merge_wait_ms is not None or merge_fill is not None:
                _lib.check(self._lib.fw_model_set_merge_wait(primary.handle, -1 if merge_wait_ms is None else int(merge_wait_ms),
                                                             90 if merge_fill is None else int(merge_fill)))
        self._seed_counter = itertools.count(1)
        self._rr = itertools.count()
        self._tls = threading.local()
        self.inter_threads = len(self._replicas)   # batches the host drivers may keep in flight

    # ---- properties read by the reference host code -------------------------------------
    @property
    def is_multilingual(self) -> bool:
        return bool(self._cfg.is_multilingual)

    @property
    def n_mels(self) -> int:
        return self._cfg.n_mels

    @property
    def device(self) -> str:
        return "cuda"

    @property
    def device_index(self) -> List[int]:
        return list(self._device_index)

    @property
    def compute_type(self) -> str:
        return self._compute_type_name

    @property
    def num_languages(self) -> int:
        return self._cfg.n_langs

    @property
    def config(self) -> WhisperConfig:
        return self._cfg

    def blob(self, replica: int = 0) -> Tuple[int, int]:
        """(device pointer, bytes) of the packed weight blob of a device's primary replica (fwamd.h: fw_model_blob) — what
        rank 0 hands to sharding.broadcast_blob_dev and what `blob_dev=` takes on the other ranks"""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self._lib.fw_model_blob(self._replicas[self._primary_of(replica)].handle, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_decode_lanes(self, lanes: int):
        """decode runs a device's group may have in flight (fwamd.h: fw_model_set_decode_lanes): 2, or 1 for per-kernel timing"""
        for i in sorted({self._primary_of(r) for r in range(len(self._replicas))}):
            _lib.check(self._lib.fw_model_set_decode_lanes(self._replicas[i].handle, int(lanes)))

    def synchronize(self):
        """block until everything queued on the encoder and decode streams of every worker has finished"""
        for r in self._replicas:
            _lib.check(self._lib.fw_synchronize(r.handle))

    def decode_stats(self) -> dict:
        """counters of the decode group of device 0: how many generate() calls shared how many decode runs"""
        runs, reqs, chunks, mx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _lib.check(self._lib.fw_model_decode_stats(self._replicas[0].handle, C.byref(runs), C.byref(reqs),
                                                   C.byref(chunks), C.byref(mx)))
        return {"runs": runs.value, "requests": reqs.value, "chunks": chunks.value, "max_run_chunks": mx.value,
                "decode_batch": int(self._lib.fw_model_decode_batch(self._replicas[0].handle)),
                "run_capacity": int(self._lib.fw_model_run_capacity(self._replicas[0].handle))}

    def dec_big_min_rows(self) -> int:
        """rows from which a decode run's linears take the GEMM-shaped kernel (bench.py prices them accordingly)"""
        return int(self._lib.fw_dec_big_min_rows())

    def dec_big_min_rows_of(self, role: int) -> int:
        """the same for one linear (0 qkv, 1 d x d, 2 ffn1, 3 ffn2) of THIS model's compute type: every linear has its own
        measured crossover (include/fwamd_test.h)"""
        if not hasattr(self._lib, "fw_dec_big_min_rows_of"):      # an older build loaded through FWAMD_LIB (A/B of two builds)
            return self.dec_big_min_rows()
        return int(self._lib.fw_dec_big_min_rows_of(int(role), 1 if self._compute_type_name == "int8_float16" else 0))

    def _replica_for(self, features: Optional[StorageView]) -> _Replica:
        if features is not None and features._owner is not None:
            return features._owner
        # thread affinity: a host thread keeps the replica it was first given (round-robin), so W
        # threads on W replicas never contend and an encoder output stays with its worker
        r = getattr(self._tls, "replica", None)
        if r is None:
            r = self._replicas[next(self._rr) % len(self._replicas)]
            self._tls.replica = r
        return r

    def _as_encoded(self, features: StorageView) -> StorageView:
        """generate/detect_language/align accept an encoder output or raw features (like CTranslate2)."""
        if features._handle is not None or features._parts:
            return features
        a = np.asarray(features._array)
        if a.ndim == 3 and a.shape[1] == self._cfg.n_audio_ctx and a.shape[2] == self._cfg.d_model:
            # encoder output that went through the host (to_cpu=True, transcribe.py:1392-1394)
            rep = self._replica_for(None)
            a = np.ascontiguousarray(a, dtype=np.float32)
            h = C.c_void_p()
            _lib.check(self._lib.fw_tensor_from_host(rep.handle, _lib.ptr(a), a.shape[0], C.byref(h)))
            return StorageView(handle=h, owner=rep, shape=a.shape)
        return self.encode(features)

    # ---- encode -------------------------------------------------------------------------
    def encode(self, features: StorageView, to_cpu: bool = False) -> StorageView:
        if not isinstance(features, StorageView):
            features = StorageView.from_array(features)
        a = features._array
        if a is None:
            raise ValueError("encode expects host features")
        if a.ndim != 3 or a.shape[1] != self._cfg.n_mels or a.shape[2] != 3000:
            raise ValueError(
                f"Invalid input features shape: expected an input with shape (B, {self._cfg.n_mels}, 3000), "
                f"but got an input with shape {tuple(a.shape)} instead")
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape[0] > self._max_batch:
            out = StorageView(parts=[self.encode(StorageView.from_array(a[i:i + self._max_batch]))
                                     for i in range(0, a.shape[0], self._max_batch)])
            return StorageView.from_array(out.to_numpy()) if to_cpu else out
        rep = self._replica_for(None)
        h = C.c_void_p()
        _lib.check(self._lib.fw_encode(rep.handle, _lib.ptr(a), a.shape[0], C.byref(h)))
        out = StorageView(handle=h, owner=rep, shape=(a.shape[0], self._cfg.n_audio_ctx, self._cfg.d_model))
        if to_cpu:
            return StorageView.from_array(out.to_numpy())
        return out

    def encode_pcm(self, chunks: Sequence[np.ndarray]) -> StorageView:
        """Fused resident path (not in CTranslate2): ragged PCM -> log-mel -> encoder on the GPU."""
        if len(chunks) > self._max_batch:
            return StorageView(parts=[self.encode_pcm(chunks[i:i + self._max_batch])
                                      for i in range(0, len(chunks), self._max_batch)])
        rep = self._replica_for(None)
        pcm, offs = _ragged(chunks, np.float32)
        h = C.c_void_p()
        _lib.check(self._lib.fw_encode_pcm(rep.handle, _lib.ptr(pcm), _lib.as_i64p(offs), len(chunks), C.byref(h)))
        return StorageView(handle=h, owner=rep, shape=(len(chunks), self._cfg.n_audio_ctx, self._cfg.d_model))

    def stage_pcm(self, chunks: Sequence[np.ndarray], replica: int = 0):
        """Copy ragged PCM chunks into HBM once; returns an opaque staged batch for encode_pcm_staged
        (bench.py: inputs are resident in HBM when the timed region starts)."""
        rep = self._replicas[replica]
        pcm, offs = _ragged(chunks, np.float32)
        dev = C.c_void_p()
        _lib.check(self._lib.fw_dev_alloc(rep.handle, pcm.nbytes, C.byref(dev)))
        _lib.check(self._lib.fw_dev_upload(rep.handle, dev, _lib.ptr(pcm), pcm.nbytes))
        return {"dev": dev, "offsets": offs, "B": len(chunks), "rep": rep}

    def free_staged(self, staged):
        _lib.check(self._lib.fw_dev_free(staged["rep"].handle, staged["dev"]))

    def encode_pcm_staged(self, staged) -> StorageView:
        # the staged PCM lives in HBM of the device: any worker replica of that device may consume it
        rep = self._replica_for(None)
        if rep.device_index != staged["rep"].device_index:
            rep = staged["rep"]
        h = C.c_void_p()
        _lib.check(self._lib.fw_encode_pcm_dev(rep.handle, staged["dev"], _lib.as_i64p(staged["offsets"]),
                                               staged["B"], C.byref(h)))
        return StorageView(handle=h, owner=rep, shape=(staged["B"], self._cfg.n_audio_ctx, self._cfg.d_model))

    def log_mel(self, chunks: Sequence[np.ndarray]) -> np.ndarray:
        """FeatureExtractor(chunk)[..., :-1] + pad_or_trim for a batch of chunks, on the GPU."""
        rep = self._replica_for(None)
        out = np.empty((len(chunks), self._cfg.n_mels, 3000), dtype=np.float32)
        for b0 in range(0, len(chunks), self._max_batch):      # the engine's workspaces hold max_batch_size chunks
            part = chunks[b0:b0 + self._max_batch]
            pcm, offs = _ragged(part, np.float32)
            _lib.check(self._lib.fw_logmel(rep.handle, _lib.ptr(pcm), _lib.as_i64p(offs), len(part),
                                           _lib.ptr(out[b0:b0 + len(part)]), None))
        return out

    def log_mel_full(self, waveform: np.ndarray) -> np.ndarray:
        rep = self._replica_for(None)
        w = np.ascontiguousarray(waveform, dtype=np.float32)
        nf = w.shape[0] // 160 + 1
        out = np.empty((self._cfg.n_mels, nf), dtype=np.float32)
        _lib.check(self._lib.fw_logmel_full(rep.handle, _lib.ptr(w), w.shape[0], _lib.ptr(out), nf))
        return out

    # ---- generate -----------------------------------------------------------------------
    def generate(self, features: StorageView, prompts: List[List[int]], *, asynchronous: bool = False,
                 beam_size: int = 5, patience: float = 1, num_hypotheses: int = 1, length_penalty: float = 1,
                 repetition_penalty: float = 1, no_repeat_ngram_size: int = 0, max_length: int = 448,
                 return_scores: bool = False, return_logits_vocab: bool = False, return_no_speech_prob: bool = False,
                 max_initial_timestamp_index: int = 50, suppress_blank: bool = True,
                 suppress_tokens: Optional[Sequence[int]] = (-1,), sampling_topk: int = 1,
                 sampling_temperature: float = 1, min_new_tokens: int = 0,
                 seed: Optional[int] = None) -> List[WhisperGenerationResult]:
        if asynchronous or return_logits_vocab:
            raise ValueError("asynchronous / return_logits_vocab are not supported (unused by faster-whisper)")
        enc = self._as_encoded(features if isinstance(features, StorageView) else StorageView.from_array(features))
        B = enc._shape[0]
        if len(prompts) != B:
            raise ValueError(f"got {len(prompts)} prompts for a batch of {B}")
        if any(len(p) == 0 for p in prompts):
            raise ValueError("prompts must not be empty")
        if enc._parts:
            kw = dict(beam_size=beam_size, patience=patience, num_hypotheses=num_hypotheses,
                      length_penalty=length_penalty, repetition_penalty=repetition_penalty,
                      no_repeat_ngram_size=no_repeat_ngram_size, max_length=max_length, return_scores=return_scores,
                      return_no_speech_prob=return_no_speech_prob,
                      max_initial_timestamp_index=max_initial_timestamp_index, suppress_blank=suppress_blank,
                      suppress_tokens=suppress_tokens, sampling_topk=sampling_topk,
                      sampling_temperature=sampling_temperature, min_new_tokens=min_new_tokens, seed=seed)
            out, b0 = [], 0
            for part in enc._parts:
                out.extend(self.generate(part, prompts[b0:b0 + part._shape[0]], **kw))
                b0 += part._shape[0]
            return out
        flat, offs = _ragged([np.asarray(p, dtype=np.int32) for p in prompts], np.int32, np.int32)
        o = _lib.FwGenOpts()
        o.beam_size, o.patience, o.num_hypotheses = int(beam_size), float(patience), int(num_hypotheses)
        o.length_penalty, o.repetition_penalty = float(length_penalty), float(repetition_penalty)
        o.no_repeat_ngram_size, o.max_length = int(no_repeat_ngram_size), int(max_length)
        o.return_scores, o.return_no_speech_prob = int(return_scores), int(return_no_speech_prob)
        o.max_initial_timestamp_index, o.suppress_blank = int(max_initial_timestamp_index), int(suppress_blank)
        sup = np.asarray([t for t in (suppress_tokens or []) if t >= 0], dtype=np.int32)
        o.suppress_tokens = _lib.as_i32p(sup) if sup.size else None
        o.n_suppress_tokens = int(sup.size)
        o.sampling_topk, o.sampling_temperature = int(sampling_topk), float(sampling_temperature)
        if seed is None:
            # CTranslate2 draws from a global generator that advances from call to call; here every sampling call
            # gets a seed of its own from a per-model counter (explicit seeds reproduce a call exactly)
            seed = call_seed(next(self._seed_counter))
        o.seed, o.min_new_tokens = int(seed), int(min_new_tokens)
        nh = max(1, int(num_hypotheses))
        ml = max(1, int(max_length))
        ids = np.zeros((B, nh, ml), dtype=np.int32)
        lens = np.zeros((B, nh), dtype=np.int32)
        scores = np.zeros((B, nh), dtype=np.float32)
        nsp = np.zeros((B,), dtype=np.float32)
        rep = enc._owner
        _lib.check(self._lib.fw_generate(rep.handle, enc._handle, _lib.as_i32p(flat), _lib.as_i32p(offs), B,
                                         C.byref(o), _lib.as_i32p(ids), _lib.as_i32p(lens), _lib.as_f32p(scores),
                                         _lib.as_f32p(nsp)))
        out = []
        for b in range(B):
            seqs = [ids[b, h, :lens[b, h]].tolist() for h in range(nh)]
            sc = [float(scores[b, h]) for h in range(nh)] if return_scores else []
            out.append(WhisperGenerationResult(seqs, sc, float(nsp[b]) if return_no_speech_prob else 0.0))
        return out

    # ---- detect_language ----------------------------------------------------------------
    def detect_language(self, features: StorageView) -> List[List[Tuple[str, float]]]:
        if not self.is_multilingual:
            raise RuntimeError("detect_language can only be called on multilingual models")
        enc = self._as_encoded(features if isinstance(features, StorageView) else StorageView.from_array(features))
        if enc._parts:
            return [row for part in enc._parts for row in self.detect_language(part)]
        B, nl = enc._shape[0], self._cfg.n_langs
        ids = np.zeros((B, nl), dtype=np.int32)
        probs = np.zeros((B, nl), dtype=np.float32)
        _lib.check(self._lib.fw_detect_language(enc._owner.handle, enc._handle, B, _lib.as_i32p(ids),
                                                _lib.as_f32p(probs)))
        names = language_token_strings(self._cfg)
        return [[(names[int(t) - self._cfg.lang_begin], float(p)) for t, p in zip(ids[b], probs[b])]
                for b in range(B)]

    # ---- align --------------------------------------------------------------------------
    def align(self, features: StorageView, start_sequence: Sequence[int], text_tokens: List[List[int]],
              num_frames: Union[int, Sequence[int]], *, median_filter_width: int = 7
              ) -> List[WhisperAlignmentResult]:
        enc = self._as_encoded(features if isinstance(features, StorageView) else StorageView.from_array(features))
        B = enc._shape[0]
        if len(text_tokens) != B:
            raise ValueError(f"got {len(text_tokens)} token lists for a batch of {B}")
        nf = np.asarray([num_frames] * B if isinstance(num_frames, int) else list(num_frames), dtype=np.int32)
        if nf.shape[0] != B:
            raise ValueError("num_frames must be an int or have one entry per batch item")
        if enc._parts:
            out, b0 = [], 0
            for part in enc._parts:
                n = part._shape[0]
                out.extend(self.align(part, start_sequence, text_tokens[b0:b0 + n], nf[b0:b0 + n].tolist(),
                                      median_filter_width=median_filter_width))
                b0 += n
            return out
        start = np.asarray(list(start_sequence), dtype=np.int32)
        flat, offs = _ragged([np.asarray(t, dtype=np.int32) for t in text_tokens], np.int32, np.int32)
        max_pairs = int(max((len(t) for t in text_tokens), default=0)) + self._cfg.n_audio_ctx + 2
        pairs = np.zeros((B, max_pairs, 2), dtype=np.int32)
        npairs = np.zeros((B,), dtype=np.int32)
        probs = np.zeros((max(1, flat.shape[0]),), dtype=np.float32)
        _lib.check(self._lib.fw_align(enc._owner.handle, enc._handle, _lib.as_i32p(start), start.shape[0],
                                      _lib.as_i32p(flat), _lib.as_i32p(offs), _lib.as_i32p(nf), B,
                                      int(median_filter_width), max_pairs, _lib.as_i32p(pairs),
                                      _lib.as_i32p(npairs), _lib.as_f32p(probs)))
        out = []
        for b in range(B):
            al = [(int(a), int(t)) for a, t in pairs[b, :npairs[b]]]
            out.append(WhisperAlignmentResult(al, probs[offs[b]:offs[b + 1]].tolist()))
        return out

    # ---- measurement --------------------------------------------------------------------
    def _primary_of(self, replica: int) -> int:
        """index of the replica that owns the decode workspace of `replica`'s device (the first one of the device)"""
        dev = self._replicas[replica].device_index
        return next(i for i, r in enumerate(self._replicas) if r.device_index == dev)

    def profile(self, enable: bool = True, replica: Optional[int] = 0):
        """per-kernel-family HIP-event timing.  replica=None: every worker.  Decode-side families are recorded by
        the replica that owns the device's decode workspace, so that one is switched along."""
        idx = range(len(self._replicas)) if replica is None else sorted({replica, self._primary_of(replica)})
        for i in idx:
            self._lib.fw_prof_enable(self._replicas[i].handle, int(enable))
            self._lib.fw_prof_reset(self._replicas[i].handle)

    def profile_report(self, replica: Optional[int] = 0) -> Dict[str, dict]:
        out = {}
        idx = range(len(self._replicas)) if replica is None else sorted({replica, self._primary_of(replica)})
        for r in idx:
            h = self._replicas[r].handle
            for i in range(self._lib.fw_prof_count()):
                ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
                _lib.check(self._lib.fw_prof_get(h, i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
                acc = out.setdefault(self._lib.fw_prof_name(i).decode(), dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                acc["ms"] += ms.value
                acc["launches"] += n.value
                acc["flops"] += fl.value
                acc["bytes"] += by.value
        return out

    def unload_model(self):
        for r in reversed(self._replicas):     # workers before the replica whose decoder / weights they share
            r.close()

    def __del__(self):
        try:
            self.unload_model()
        except Exception:
            pass


_LANG_CODES = (
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi "
    "ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd "
    "gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su yue").split()


def language_token_strings(cfg: WhisperConfig) -> List[str]:
    """'<|en|>' ... in vocabulary order (the reference strips the markers, transcribe.py:1826)."""
    codes = _LANG_CODES[:cfg.n_langs] if cfg.n_langs <= len(_LANG_CODES) else \
        _LANG_CODES + [f"l{i}" for i in range(len(_LANG_CODES), cfg.n_langs)]
    return [f"<|{c}|>" for c in codes]


def _ragged(seqs, dtype, off_dtype=np.int64):
    offs = np.zeros(len(seqs) + 1, dtype=off_dtype)
    for i, s in enumerate(seqs):
        offs[i + 1] = offs[i] + len(s)
    flat = np.zeros(max(1, int(offs[-1])), dtype=dtype)
    for i, s in enumerate(seqs):
        flat[offs[i]:offs[i + 1]] = np.asarray(s, dtype=dtype)
    return flat, offs
