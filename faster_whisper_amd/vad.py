"""Host-side speech chunking of the batched path (next-row f-3 for the network).

Mirrors the reference's `faster_whisper/vad.py` interface (same names, argument meaning, return layout):
    VadOptions                      vad.py:14-43
    get_speech_timestamps           vad.py:46-183   hysteresis state machine over per-window speech probabilities
    collect_chunks                  vad.py:186-243  merge speech spans into <= max_duration chunks
    SpeechTimestampsMap             vad.py:246-285  map times on the silence-free axis back to the recording

    SileroVADModel, get_vad_model   vad.py:288-351  the Silero VAD v6 network (row f-3)

The network that produces the window probabilities is an ONNX asset the reference runs with onnxruntime on one
CPU thread.  Here it is native host C++ behind the C ABI (`fw_vad_*`, csrc/vad_host.cpp: window-parallel front
end on a thread pool, sequential LSTM recurrence); the ONNX file is only read for its weights (onnx_lite.py) and
is NOT redistributed: `get_vad_model()` looks at $FWAMD_SILERO_VAD_ONNX, then at an installed `faster_whisper`
package, and fails loudly otherwise.  `get_speech_timestamps` also accepts the probabilities directly
(`speech_probs=`) or any callable with SileroVADModel's contract (`vad_model=`).  Parity of the network with the
reference's onnxruntime run is unpinned (onnxruntime absent); the state machine downstream is pinned
(tests/golden/host_units.json).
"""
import bisect
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

WINDOW = 512   # samples per VAD window at 16 kHz (vad.py:70)


@dataclass
class VadOptions:
    """threshold: probability at/above which a window is speech; neg_threshold: below it a triggered span starts
    counting silence (default threshold - 0.15, floor 0.01); min_speech_duration_ms: shorter spans are dropped;
    max_speech_duration_s: longer spans are cut at the last >98 ms silence (or hard); min_silence_duration_ms:
    silence needed to close a span; speech_pad_ms: padding on both sides of every span."""
    threshold: float = 0.5
    neg_threshold: Optional[float] = None
    min_speech_duration_ms: int = 0
    max_speech_duration_s: float = float("inf")
    min_silence_duration_ms: int = 2000
    speech_pad_ms: int = 400


def get_speech_timestamps(audio: np.ndarray, vad_options: Optional[VadOptions] = None, sampling_rate: int = 16000,
                          speech_probs: Optional[Sequence[float]] = None,
                          vad_model: Optional[Callable[[np.ndarray], np.ndarray]] = None, _every_window: bool = False,
                          **kwargs) -> List[dict]:
    """-> [{"start": sample, "end": sample}, ...] speech spans of `audio` (1-D float array).
    _every_window (tests): walk every window like the reference instead of jumping over the ones that cannot change the
    state — the two must return the same spans."""
    opts = vad_options if vad_options is not None else VadOptions(**kwargs)
    n_audio = len(audio)
    if speech_probs is None:
        if vad_model is None:
            vad_model = get_vad_model()
        # the reference always appends 1..512 zero samples (a whole extra window when already aligned)
        padded = np.pad(audio, (0, WINDOW - n_audio % WINDOW))
        speech_probs = vad_model(padded)
    probs = np.asarray(speech_probs, dtype=np.float64).reshape(-1)

    thr = opts.threshold
    neg = opts.neg_threshold if opts.neg_threshold is not None else max(thr - 0.15, 0.01)
    per_ms = sampling_rate / 1000
    min_speech = per_ms * opts.min_speech_duration_ms
    pad = per_ms * opts.speech_pad_ms
    max_speech = sampling_rate * opts.max_speech_duration_s - WINDOW - 2 * pad
    min_silence = per_ms * opts.min_silence_duration_ms
    min_silence_at_max = per_ms * 98

    spans: List[dict] = []
    start = None          # start sample of the open span (None: not triggered)
    silence_at = 0        # sample where the current run of sub-neg windows began (0: none)
    cut_end = cut_next = 0   # candidate cut (end of a >98 ms silence, restart point) for over-long spans

    def close(end):
        nonlocal start, silence_at, cut_end, cut_next
        spans.append({"start": start, "end": end})
        start, silence_at, cut_end, cut_next = None, 0, 0, 0

    # The reference walks every window (vad.py:85-160).  Most windows change nothing: outside a span only a window with
    # p >= thr does, inside a span with no silence run open only a window with p < neg or the over-long test does.  The
    # loop below runs the SAME body on exactly the windows that can change the state and jumps over the rest (an 8 h
    # recording is 900 000 windows: 1.3 s of Python per call, in front of the first transcribed chunk);
    # `_every_window=True` keeps the plain walk and tests/test_host_logic.py checks the two against each other.
    n = probs.shape[0]
    at_thr = np.flatnonzero(probs >= thr)          # windows that can open a span / end a silence run
    below = np.flatnonzero(probs < neg)            # windows that can open a silence run
    i = 0
    while i < n:
        if _every_window:
            pass
        elif start is None:
            k = np.searchsorted(at_thr, i)
            if k == at_thr.shape[0]:
                break
            i = int(at_thr[k])
        elif not silence_at:
            # nothing happens before the next sub-neg window or the first window with pos - start > max_speech
            k = np.searchsorted(below, i)
            nxt = int(below[k]) if k < below.shape[0] else n
            over = n if max_speech == float("inf") else int((start + max_speech) // WINDOW) + 1
            i = max(i, min(nxt, over))
            if i >= n:
                break
        p = probs[i]
        pos = WINDOW * i
        i += 1
        if p >= thr and silence_at:
            silence_at = 0
            if cut_next < cut_end:
                cut_next = pos
        if p >= thr and start is None:
            start = pos
            continue
        if start is not None and pos - start > max_speech:
            if cut_end:
                resume = cut_next if cut_next >= cut_end else None
                close(cut_end)
                start = resume
            else:
                close(pos)
                continue
        if p < neg and start is not None:
            if not silence_at:
                silence_at = pos
            if pos - silence_at > min_silence_at_max:
                cut_end = silence_at
            if pos - silence_at < min_silence:
                continue
            if silence_at - start > min_speech:
                close(silence_at)
            else:
                start, silence_at, cut_end, cut_next = None, 0, 0, 0
    if start is not None and n_audio - start > min_speech:
        spans.append({"start": start, "end": n_audio})

    # padding: split the gap between neighbours when it is shorter than two pads
    for k, span in enumerate(spans):
        if k == 0:
            span["start"] = int(max(0, span["start"] - pad))
        if k + 1 < len(spans):
            nxt = spans[k + 1]
            gap = nxt["start"] - span["end"]
            if gap < 2 * pad:
                span["end"] += int(gap // 2)
                nxt["start"] = int(max(0, nxt["start"] - gap // 2))
            else:
                span["end"] = int(min(n_audio, span["end"] + pad))
                nxt["start"] = int(max(0, nxt["start"] - pad))
        else:
            span["end"] = int(min(n_audio, span["end"] + pad))
    return spans


def collect_chunks(audio: np.ndarray, chunks: List[dict], sampling_rate: int = 16000,
                   max_duration: float = float("inf")) -> Tuple[List[np.ndarray], List[Dict[str, float]]]:
    """Concatenates speech spans into chunks of at most `max_duration` seconds.
    -> (audio per chunk, metadata per chunk {"offset": s on the silence-free axis, "duration": s, "segments": spans})"""
    if not chunks:
        return [np.array([], dtype=np.float32)], [{"offset": 0, "duration": 0, "segments": []}]
    limit = max_duration * sampling_rate
    out_audio, out_meta = [], []
    pieces: List[np.ndarray] = []
    members: List[dict] = []
    length = 0       # samples in the open chunk
    emitted = 0      # samples in the chunks already emitted

    def flush():
        nonlocal emitted
        # (one span: the slice itself — a contiguous view of the recording, no copy; an 8 h recording of single-span chunks
        #  is otherwise 1.8 GB of memcpy in front of the first transcribed chunk)
        out_audio.append(pieces[0] if len(pieces) == 1 else
                         (np.concatenate(pieces) if pieces else np.array([], dtype=np.float32)))
        out_meta.append({"offset": emitted / sampling_rate, "duration": length / sampling_rate, "segments": members})
        emitted += length

    for span in chunks:
        n = span["end"] - span["start"]
        if length + n > limit:
            flush()
            # (reference behaviour, vad.py:212-227: the span that opens a new chunk is not listed in "segments")
            pieces, members, length = [audio[span["start"]:span["end"]]], [], n
        else:
            pieces.append(audio[span["start"]:span["end"]])
            members.append(span)
            length += n
    flush()
    return out_audio, out_meta


class SpeechTimestampsMap:
    """Restores times measured on the concatenated-speech axis to the original recording."""

    def __init__(self, chunks: List[dict], sampling_rate: int, time_precision: int = 2):
        self.sampling_rate = sampling_rate
        self.time_precision = time_precision
        self.chunk_end_sample: List[int] = []
        self.total_silence_before: List[float] = []
        removed, prev_end = 0, 0
        for c in chunks:
            removed += c["start"] - prev_end
            prev_end = c["end"]
            self.chunk_end_sample.append(c["end"] - removed)
            self.total_silence_before.append(removed / sampling_rate)

    def get_chunk_index(self, time: float, is_end: bool = False) -> int:
        sample = int(time * self.sampling_rate)
        if is_end and sample in self.chunk_end_sample:
            return self.chunk_end_sample.index(sample)
        return min(bisect.bisect(self.chunk_end_sample, sample), len(self.chunk_end_sample) - 1)

    def get_original_time(self, time: float, chunk_index: Optional[int] = None, is_end: bool = False) -> float:
        if chunk_index is None:
            chunk_index = self.get_chunk_index(time, is_end)
        return round(self.total_silence_before[chunk_index] + time, self.time_precision)


# ---- the Silero VAD v6 network (row f-3): native host implementation behind the C ABI ---------------------
_VAD_MODEL = None
ONNX_ENV = "FWAMD_SILERO_VAD_ONNX"


def find_vad_onnx() -> Optional[str]:
    """Where the network's weights come from: the ONNX asset the reference ships
    (`faster_whisper/assets/silero_vad_v6.onnx`, vad.py:288-292).  This repository does not redistribute it:
    $FWAMD_SILERO_VAD_ONNX, else the assets directory of an installed `faster_whisper` package."""
    import importlib.util
    import os
    path = os.environ.get(ONNX_ENV)
    if path:
        return path if os.path.isfile(path) else None
    try:
        spec = importlib.util.find_spec("faster_whisper")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "assets", "silero_vad_v6.onnx")
        if os.path.isfile(cand):
            return cand
    return None


def get_vad_model():
    """Cached SileroVADModel (vad.py:288-292); fails loudly when the weights cannot be found."""
    global _VAD_MODEL
    if _VAD_MODEL is None:
        path = find_vad_onnx()
        if path is None:
            raise RuntimeError(
                "Silero VAD weights not found: set FWAMD_SILERO_VAD_ONNX to the reference's "
                "faster_whisper/assets/silero_vad_v6.onnx (or install faster_whisper), or pass speech_probs= / "
                "vad_model= / clip_timestamps")
        _VAD_MODEL = SileroVADModel(path)
    return _VAD_MODEL


class SileroVADModel:
    """Same call contract as the reference's SileroVADModel (vad.py:295-351): `model(padded_audio)` -> one speech
    probability per 512-sample window.  The network runs in libfwamd.so's host C++ implementation
    (csrc/vad.hip, `fw_vad_*`); the ONNX file is only the container of the weights (onnx_lite.py)."""

    SHAPES = {"encoder.feature_extractor.forward_basis_buffer": (258, 1, 256),
              "encoder.conv_layers.0.weight": (128, 129, 3), "encoder.conv_layers.1.weight": (64, 128, 3),
              "encoder.conv_layers.2.weight": (64, 64, 3), "encoder.conv_layers.3.weight": (128, 64, 3),
              "decoder.conv1d.weight": (1, 128, 1)}

    def __init__(self, path: Optional[str] = None, weights: Optional[Dict[str, np.ndarray]] = None,
                 n_threads: int = 0, device: str = "cpu", device_index: int = 0):
        """device="cpu" (default): host C++ on `n_threads` threads (0 = all cores).  device="cuda": the HIP kernels
        of csrc/vad.hip — opt-in until they have been validated on hardware."""
        import ctypes as C
        from . import _lib, onnx_lite
        if weights is None:
            _, weights, _, _ = onnx_lite.load(path)
        for name, shape in self.SHAPES.items():
            if name not in weights or tuple(weights[name].shape) != shape:
                raise ValueError(f"not a Silero VAD v6 model: initializer '{name}' missing or not {shape}")
        mats = [v for v in weights.values() if tuple(v.shape) == (1, 512, 128)]
        bias = [v for v in weights.values() if tuple(v.shape) == (1, 1024)]
        if len(mats) != 2 or len(bias) != 1:
            raise ValueError("not a Silero VAD v6 model: expected LSTM W, R [1,512,128] and B [1,1024]")
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)   # noqa: E731
        self._keep = dict(
            basis=f32(weights["encoder.feature_extractor.forward_basis_buffer"]),
            cw=[f32(weights[f"encoder.conv_layers.{i}.weight"]) for i in range(4)],
            cb=[f32(weights[f"encoder.conv_layers.{i}.bias"]) for i in range(4)],
            lw=f32(mats[0]), lr=f32(mats[1]), lb=f32(bias[0]), dw=f32(weights["decoder.conv1d.weight"]))
        k = self._keep
        w = _lib.FwVadWeights()
        w.stft_basis = k["basis"].ctypes.data
        for i in range(4):
            w.conv_w[i] = k["cw"][i].ctypes.data
            w.conv_b[i] = k["cb"][i].ctypes.data
        w.lstm_w, w.lstm_r, w.lstm_b = k["lw"].ctypes.data, k["lr"].ctypes.data, k["lb"].ctypes.data
        w.dec_w = k["dw"].ctypes.data
        w.dec_b = float(np.asarray(weights["decoder.conv1d.bias"]).reshape(-1)[0])
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        _lib.check(self._lib.fw_vad_create(C.byref(w), C.byref(self._handle)))
        self.n_threads = n_threads
        if device not in ("cpu", "cuda"):
            raise ValueError(f"unsupported device '{device}'")
        self.device, self.device_index = device, device_index

    def __call__(self, audio: np.ndarray, num_samples: int = 512, context_size_samples: int = 64) -> np.ndarray:
        from . import _lib
        assert audio.ndim == 1, "Input should be a 1D array"
        assert audio.shape[0] % num_samples == 0, "Input size should be a multiple of num_samples"
        if (num_samples, context_size_samples) != (512, 64):
            raise ValueError("the Silero v6 network takes 512-sample windows with 64 samples of context")
        h = np.zeros(128, dtype=np.float32)
        c = np.zeros(128, dtype=np.float32)
        if self.device == "cuda" and audio.shape[0] > 0:
            # the device frames the recording itself (fw_vad_forward_audio_dev): no [n][576] host copy, 12 % fewer bytes
            # over PCIe — for an 8 h recording the numpy framing below alone was most of a second
            a = np.ascontiguousarray(audio, dtype=np.float32)
            n = a.shape[0] // num_samples
            probs = np.empty(n, dtype=np.float32)
            _lib.check(self._lib.fw_vad_forward_audio_dev(self._handle, self.device_index, _lib.ptr(a), a.shape[0],
                                                          _lib.ptr(h), _lib.ptr(c), _lib.ptr(probs)))
            return probs
        # framing of the reference (vad.py:318-336): the context of a window is the tail of the previous one,
        # zeros for the first; its in-place `context[-1] = 0` also clears the tail of the last (padding) window
        win = np.array(audio, dtype=np.float32).reshape(-1, num_samples)
        win[-1, -context_size_samples:] = 0
        ctx = np.roll(win[:, -context_size_samples:], 1, axis=0)
        windows = np.ascontiguousarray(np.concatenate([ctx, win], axis=1))
        n = windows.shape[0]
        probs = np.empty(n, dtype=np.float32)
        if self.device == "cuda":
            _lib.check(self._lib.fw_vad_forward_dev(self._handle, self.device_index, _lib.ptr(windows), n, _lib.ptr(h),
                                                    _lib.ptr(c), _lib.ptr(probs)))
        else:
            _lib.check(self._lib.fw_vad_forward(self._handle, _lib.ptr(windows), n, self.n_threads, _lib.ptr(h),
                                                _lib.ptr(c), _lib.ptr(probs)))
        return probs

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                self._lib.fw_vad_free(h)
            except Exception:
                pass
