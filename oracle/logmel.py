"""numpy restatement of the reference log-mel front end (test oracle).

Follows /root/reference/faster_whisper/feature_extractor.py:
  get_mel_filters :25-65, stft :68-196 (centre reflect pad :117-121, framing :157-168,
  window :170-171, rfft :189), __call__ :198-230,
and the batched driver's framing: `[..., :-1]` (transcribe.py:464) + pad_or_trim to 3000
frames with zeros (audio.py:111-123, transcribe.py:515).

Written independently of the reference code (explicit gather framing instead of
as_strided, explicit triangular reflect index instead of np.pad); pinned against the
reference by tests/test_oracle_logmel.py using tests/golden/logmel_*.npz.
"""
import numpy as np

N_FFT = 400
HOP = 160
SR = 16000
N_FRAMES = 3000


def mel_filters(n_mels: int) -> np.ndarray:
    """Slaney-normalised triangular filterbank, float64 math, float32 result [n_mels, 201]."""
    fft_hz = np.arange(N_FFT // 2 + 1, dtype=np.float64) * (1.0 / (N_FFT * (1.0 / SR)))
    mel_pts = np.linspace(0.0, 45.245640471924965, n_mels + 2)
    hz = (200.0 / 3) * mel_pts
    brk_mel = 1000.0 / (200.0 / 3)
    logstep = np.log(6.4) / 27.0
    up = mel_pts >= brk_mel
    hz[up] = 1000.0 * np.exp(logstep * (mel_pts[up] - brk_mel))
    out = np.zeros((n_mels, fft_hz.size), dtype=np.float64)
    for i in range(n_mels):
        lo, ce, hi = hz[i], hz[i + 1], hz[i + 2]
        rise = (fft_hz - lo) / (ce - lo)
        fall = (hi - fft_hz) / (hi - ce)
        out[i] = np.maximum(0.0, np.minimum(rise, fall)) * (2.0 / (hi - lo))
    return out.astype(np.float32)


def hann_window() -> np.ndarray:
    n = np.arange(N_FFT, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(np.float32)


def _reflect_index(p: np.ndarray, length: int) -> np.ndarray:
    """index into an array of `length` for positions p (may be <0 or >=length), numpy 'reflect' rule"""
    if length == 1:
        return np.zeros_like(p)
    period = 2 * (length - 1)
    q = np.mod(p, period)
    return np.where(q >= length, period - q, q)


def log_mel_full(wave: np.ndarray, n_mels: int, filters: np.ndarray = None) -> np.ndarray:
    """FeatureExtractor.__call__(wave): float32 [n_mels, len(wave)//160 + 1]."""
    wave = np.asarray(wave, dtype=np.float32)
    if filters is None:
        filters = mel_filters(n_mels)
    x = np.concatenate([wave, np.zeros(HOP, dtype=np.float32)])       # padding=160
    L = x.shape[0]
    n_frames_stft = 1 + (L + N_FFT - N_FFT) // HOP                    # after centre padding by 200+200
    pos = (np.arange(n_frames_stft)[:, None] * HOP + np.arange(N_FFT)[None, :]) - N_FFT // 2
    frames = x[_reflect_index(pos, L)] * hann_window()[None, :]       # float32 product
    spec = np.fft.rfft(frames.astype(np.float32), n=N_FFT, axis=-1).astype(np.complex64)
    power = (np.abs(spec[:-1]) ** 2).T                                # drop last frame -> [201, L//160]
    mel = filters @ power
    logm = np.log10(np.clip(mel, 1e-10, None))
    logm = np.maximum(logm, logm.max() - 8.0)
    return ((logm + 4.0) / 4.0).astype(np.float32)


def pad_or_trim(feat: np.ndarray, length: int = N_FRAMES) -> np.ndarray:
    if feat.shape[-1] > length:
        feat = feat[..., :length]
    if feat.shape[-1] < length:
        feat = np.concatenate([feat, np.zeros(feat.shape[:-1] + (length - feat.shape[-1],), feat.dtype)], axis=-1)
    return feat


def log_mel_chunks(chunks, n_mels: int) -> np.ndarray:
    """The batched driver's features: per chunk log_mel_full(chunk)[..., :-1], padded to 3000 frames.
    Returns float32 [B, n_mels, 3000]."""
    filt = mel_filters(n_mels)
    return np.stack([pad_or_trim(log_mel_full(c, n_mels, filt)[..., :-1]) for c in chunks])
