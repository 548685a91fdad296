"""Host-side mirror of the reference's batched transcription API for the hot path.

Mirrors (same names, argument meaning, defaults and error behaviour)
    faster_whisper/transcribe.py : Word, Segment, TranscriptionOptions, TranscriptionInfo (:31-108),
        BatchedInferencePipeline.forward / generate_segment_batched / transcribe /
        _batched_segments_generator (:111-617), WhisperModel.__init__ attributes (:620-722),
        encode (:1391-1400), get_prompt (:1532-1565), _split_segments_by_timestamps (:1024-1101),
        find_alignment (:1698-1766), detect_language (:1768-1841), get_suppressed_tokens (:1884-1907),
        get_compression_ratio (:1879-1881)
    faster_whisper/audio.py : pad_or_trim (:111-123)
    faster_whisper/feature_extractor.py : FeatureExtractor (the arithmetic runs on the GPU here)
but is written for this engine: features are computed on the GPU, the backend is
`faster_whisper_amd.backend.Whisper`, and `transcribe(..., shard=True)` partitions the chunk
list over the ranks of a torch.distributed job.

Also here: the sequential seek loop
(`WhisperModel.transcribe` / `generate_segments` :747-1022, :1103-1389), the temperature-fallback ladder
(`generate_with_fallback` :1402-1530) and word timestamps (`add_word_timestamps` :1567-1696; heuristics in
words.py, speech chunking in vad.py).  The host logic is pinned against the reference's own code driven by a
scripted backend (oracle/gen_golden_host.py -> tests/golden/host_*.json).

Out of scope: the Silero VAD network itself (row f-3: speech probabilities are an input
here) and PyAV decoding (row f-4: audio is a 16 kHz float32 ndarray).
"""
import json
import logging
import os
import zlib
from dataclasses import dataclass
from math import ceil
from typing import Iterable, List, Optional, Tuple, Union

import numpy as np

from .backend import StorageView, Whisper, language_token_strings
from .config import WhisperConfig

_LOG = logging.getLogger("faster_whisper")


@dataclass
class Word:
    start: float
    end: float
    word: str
    probability: float


@dataclass
class Segment:
    id: int
    seek: int
    start: float
    end: float
    text: str
    tokens: List[int]
    avg_logprob: float
    compression_ratio: float
    no_speech_prob: float
    words: Optional[List[Word]]
    temperature: Optional[float]


@dataclass
class TranscriptionOptions:
    beam_size: int
    best_of: int
    patience: float
    length_penalty: float
    repetition_penalty: float
    no_repeat_ngram_size: int
    log_prob_threshold: Optional[float]
    no_speech_threshold: Optional[float]
    compression_ratio_threshold: Optional[float]
    condition_on_previous_text: bool
    prompt_reset_on_temperature: float
    temperatures: List[float]
    initial_prompt: Optional[Union[str, Iterable[int]]]
    prefix: Optional[str]
    suppress_blank: bool
    suppress_tokens: Optional[List[int]]
    without_timestamps: bool
    max_initial_timestamp: float
    word_timestamps: bool
    prepend_punctuations: str
    append_punctuations: str
    multilingual: bool
    max_new_tokens: Optional[int]
    clip_timestamps: Union[str, List[float]]
    hallucination_silence_threshold: Optional[float]
    hotwords: Optional[str]


@dataclass
class TranscriptionInfo:
    language: str
    language_probability: float
    duration: float
    duration_after_vad: float
    all_language_probs: Optional[List[Tuple[str, float]]]
    transcription_options: TranscriptionOptions
    vad_options: object


def pad_or_trim(array: np.ndarray, length: int = 3000, *, axis: int = -1) -> np.ndarray:
    """trim to `length` frames or right-pad with zeros (audio.py:111-123)"""
    if array.shape[axis] > length:
        array = np.take(array, np.arange(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array


def get_compression_ratio(text: str) -> float:
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


class FeatureExtractor:
    """Same constructor / call signature as the reference's numpy FeatureExtractor; the
    arithmetic runs in the HIP log-mel kernel of the model it is attached to."""

    def __init__(self, feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400,
                 backend: Optional[Whisper] = None):
        if (sampling_rate, hop_length, n_fft) != (16000, 160, 400):
            raise ValueError("the HIP log-mel kernel is built for sampling_rate=16000, hop_length=160, n_fft=400")
        self.n_fft, self.hop_length, self.chunk_length = n_fft, hop_length, chunk_length
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.time_per_frame = hop_length / sampling_rate
        self.sampling_rate = sampling_rate
        self.feature_size = feature_size
        self._backend = backend

    def __call__(self, waveform: np.ndarray, padding=160, chunk_length=None) -> np.ndarray:
        if self._backend is None:
            raise RuntimeError("FeatureExtractor is not attached to a GPU model (no CPU implementation)")
        if padding != 160:
            raise ValueError("only padding=160 (the reference default) is supported")
        if chunk_length is not None:
            self.n_samples = chunk_length * self.sampling_rate
            self.nb_max_frames = self.n_samples // self.hop_length
        return self._backend.log_mel_full(np.asarray(waveform, dtype=np.float32))


class Tokenizer:
    """Special-token view of the Whisper vocabulary (tokenizer.py:9-112).  Text encode/decode
    is delegated to a HF `tokenizers.Tokenizer` when a tokenizer.json is available; without
    one (synthetic weights) ids are rendered as `<id>` so the pipeline still runs end to end."""

    def __init__(self, hf_tokenizer, cfg: WhisperConfig, multilingual: bool, task: Optional[str] = None,
                 language: Optional[str] = None):
        self.tokenizer = hf_tokenizer
        self.cfg = cfg
        self.sot, self.eot = cfg.sot, cfg.eot
        self.sot_prev, self.sot_lm = cfg.sot_prev, cfg.sot_lm
        self.no_speech, self.no_timestamps = cfg.no_speech, cfg.no_timestamps
        self.transcribe, self.translate = cfg.transcribe, cfg.translate
        self.timestamp_begin = cfg.timestamp_begin
        if multilingual:
            if task not in ("transcribe", "translate"):
                raise ValueError(f"'{task}' is not a valid task (accepted tasks: transcribe, translate)")
            names = [s[2:-2] for s in language_token_strings(cfg)]
            if language not in names:
                raise ValueError(f"'{language}' is not a valid language code")
            self.task = cfg.transcribe if task == "transcribe" else cfg.translate
            self.language = cfg.lang_begin + names.index(language)
            self.language_code = language
        else:
            self.task = None
            self.language = None
            self.language_code = "en"

    @property
    def sot_sequence(self) -> List[int]:
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.language)
        if self.task is not None:
            seq.append(self.task)
        return seq

    @property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        # needs the real vocabulary (tokenizer.py:114-148); empty for synthetic models
        if self.tokenizer is None:
            return ()
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』') + "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for symbol in symbols + list(miscellaneous):
            for tokens in (self.encode(symbol), self.encode(" " + symbol)):
                if len(tokens) == 1 or symbol in miscellaneous:
                    result.add(tokens[0])
        return tuple(sorted(result))

    def encode(self, text: str) -> List[int]:
        if self.tokenizer is None:
            raise RuntimeError("text prompts need a tokenizer.json (this model has none)")
        return self.tokenizer.encode(text, add_special_tokens=False).ids

    def decode(self, tokens: List[int]) -> str:
        text_tokens = [t for t in tokens if t < self.eot]
        return self._render(text_tokens)

    def _render(self, ids: List[int]) -> str:
        """text of a run of non-timestamp ids (special tokens are dropped, as HF `decode` does)"""
        if self.tokenizer is None:
            return "".join(f"<{t}>" for t in ids if t < self.eot)
        return self.tokenizer.decode(ids)

    def decode_with_timestamps(self, tokens: List[int]) -> str:
        """text with every timestamp token rendered as <|s.ss|> (tokenizer.py:95-112)"""
        out, run = [], []
        for t in tokens:
            if t >= self.timestamp_begin:
                out.append(self._render(run))
                out.append(f"<|{(t - self.timestamp_begin) * 0.02:.2f}|>")
                run = []
            else:
                run.append(t)
        out.append(self._render(run))
        return "".join(out)

    # ---- word splitting (tokenizer.py:150-208) -------------------------------------------
    def split_to_word_tokens(self, tokens: List[int]) -> Tuple[List[str], List[List[int]]]:
        # languages written without spaces are split wherever the bytes so far decode to whole characters
        if self.language_code in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)

    def split_tokens_on_unicode(self, tokens: List[int]) -> Tuple[List[str], List[List[int]]]:
        """groups tokens until they decode without a dangling partial UTF-8 sequence (U+FFFD), unless the
        full text really has U+FFFD at that position"""
        full = self.decode_with_timestamps(tokens)
        words, groups, pending, consumed = [], [], [], 0
        for t in tokens:
            pending.append(t)
            text = self.decode_with_timestamps(pending)
            bad = text.find("�")
            if bad < 0 or (consumed + bad < len(full) and full[consumed + bad] == "�"):
                words.append(text)
                groups.append(pending)
                consumed += len(text)
                pending = []
        return words, groups

    def split_tokens_on_spaces(self, tokens: List[int]) -> Tuple[List[str], List[List[int]]]:
        """unicode pieces are joined into words: a piece opens a new word when it is a special token, starts
        with a space, is a lone punctuation mark, or is the very first piece"""
        import string
        words: List[str] = []
        groups: List[List[int]] = []
        for piece, ids in zip(*self.split_tokens_on_unicode(tokens)):
            opens = (ids[0] >= self.eot or piece.startswith(" ") or piece.strip() in string.punctuation
                     or not words)
            if opens:
                words.append(piece)
                groups.append(ids)
            else:
                words[-1] += piece
                groups[-1].extend(ids)
        return words, groups


def get_suppressed_tokens(tokenizer: Tokenizer, suppress_tokens) -> Optional[Tuple[int, ...]]:
    """suppress set = user ids (-1 expands to the non-speech tokens) + task/sot/no_speech specials,
    sorted tuple (transcribe.py:1884-1907)"""
    if -1 in suppress_tokens:
        suppress_tokens = [t for t in suppress_tokens if t >= 0]
        suppress_tokens.extend(tokenizer.non_speech_tokens)
    elif suppress_tokens is None or len(suppress_tokens) == 0:
        suppress_tokens = []
    else:
        assert isinstance(suppress_tokens, list), "suppress_tokens must be a list"
        suppress_tokens = list(suppress_tokens)
    suppress_tokens.extend([tokenizer.transcribe, tokenizer.translate, tokenizer.sot, tokenizer.sot_prev,
                            tokenizer.sot_lm, tokenizer.no_speech])
    return tuple(sorted(set(suppress_tokens)))


class WhisperModel:
    """Model + the helpers the batched pipeline needs (reference: WhisperModel.__init__ :620-722)."""

    def __init__(self, model_size_or_path: str, device: str = "auto", device_index: Union[int, List[int]] = 0,
                 compute_type: str = "default", cpu_threads: int = 0, num_workers: int = 1,
                 download_root: Optional[str] = None, local_files_only: bool = False, files: dict = None,
                 revision: Optional[str] = None, use_auth_token=None, **model_kwargs):
        self.logger = _LOG
        tokenizer_bytes = preprocessor_bytes = None
        if files:
            files = dict(files)
            tokenizer_bytes = files.pop("tokenizer.json", None)
            preprocessor_bytes = files.pop("preprocessor_config.json", None)
        model_path = model_size_or_path
        self.model = Whisper(model_path, device=device, device_index=device_index, compute_type=compute_type,
                             intra_threads=cpu_threads, inter_threads=num_workers, files=files, **model_kwargs)
        cfg = self.model.config
        self.hf_tokenizer = None
        tok_file = os.path.join(model_path, "tokenizer.json") if isinstance(model_path, str) else ""
        if tokenizer_bytes or os.path.isfile(tok_file):
            import tokenizers
            self.hf_tokenizer = (tokenizers.Tokenizer.from_buffer(tokenizer_bytes) if tokenizer_bytes
                                 else tokenizers.Tokenizer.from_file(tok_file))
        self.feat_kwargs = self._get_feature_kwargs(model_path, preprocessor_bytes)
        self.feat_kwargs.setdefault("feature_size", cfg.n_mels)
        self.feature_extractor = FeatureExtractor(**self.feat_kwargs, backend=self.model)
        self.input_stride = 2
        self.num_samples_per_token = self.feature_extractor.hop_length * self.input_stride
        self.frames_per_second = self.feature_extractor.sampling_rate // self.feature_extractor.hop_length
        self.tokens_per_second = self.feature_extractor.sampling_rate // self.num_samples_per_token
        self.time_precision = 0.02
        self.max_length = 448

    @property
    def supported_languages(self) -> List[str]:
        if not self.model.is_multilingual:
            return ["en"]
        return [s[2:-2] for s in language_token_strings(self.model.config)]

    def _get_feature_kwargs(self, model_path, preprocessor_bytes=None) -> dict:
        config = {}
        try:
            path = os.path.join(model_path, "preprocessor_config.json") if isinstance(model_path, str) else ""
            if preprocessor_bytes:
                config = json.loads(preprocessor_bytes)
            elif os.path.isfile(path):
                with open(path, "r", encoding="utf-8") as f:
                    config = json.load(f)
            else:
                return config
            valid = ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft")
            return {k: v for k, v in config.items() if k in valid}
        except json.JSONDecodeError as e:
            self.logger.warning("Could not load preprocessor config: %s", e)
        return config

    # ---- sequential path (reference transcribe.py:747-1022, :1103-1389) ----
    def transcribe(self, audio: np.ndarray, language: Optional[str] = None, task: str = "transcribe",
                   log_progress: bool = False, beam_size: int = 5, best_of: int = 5, patience: float = 1,
                   length_penalty: float = 1, repetition_penalty: float = 1, no_repeat_ngram_size: int = 0,
                   temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), compression_ratio_threshold: Optional[float] = 2.4,
                   log_prob_threshold: Optional[float] = -1.0, no_speech_threshold: Optional[float] = 0.6,
                   condition_on_previous_text: bool = True, prompt_reset_on_temperature: float = 0.5,
                   initial_prompt=None, prefix: Optional[str] = None, suppress_blank: bool = True,
                   suppress_tokens: Optional[List[int]] = [-1], without_timestamps: bool = False,
                   max_initial_timestamp: float = 1.0, word_timestamps: bool = False,
                   prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
                   multilingual: bool = False, vad_filter: bool = False, vad_parameters=None,
                   max_new_tokens: Optional[int] = None, chunk_length: Optional[int] = None,
                   clip_timestamps: Union[str, List[float]] = "0",
                   hallucination_silence_threshold: Optional[float] = None, hotwords: Optional[str] = None,
                   language_detection_threshold: Optional[float] = 0.5, language_detection_segments: int = 1,
                   vad_speech_probs=None):
        """Sequential (seek-loop) transcription: 30 s windows decoded one after the other, each conditioned on
        the text before it, with the temperature-fallback ladder.  Same arguments / defaults / return value
        as the reference's WhisperModel.transcribe; `vad_speech_probs` (not in the reference) feeds
        vad.get_speech_timestamps with precomputed window probabilities instead of running the Silero network."""
        from .vad import VadOptions, collect_chunks, get_speech_timestamps
        from .words import restore_speech_timestamps
        sr = self.feature_extractor.sampling_rate
        if multilingual and not self.model.is_multilingual:
            self.logger.warning("The current model is English-only but the multilingual parameter is set to"
                                "True; setting to False instead.")
            multilingual = False
        if not isinstance(audio, np.ndarray):
            from .audio import decode_audio   # path / file object: WAVE and FLAC natively, other containers through PyAV
            audio = decode_audio(audio, sampling_rate=sr)
        duration = audio.shape[0] / sr
        duration_after_vad = duration
        speech_chunks = None
        if vad_filter and clip_timestamps == "0":
            if vad_parameters is None:
                vad_parameters = VadOptions()
            elif isinstance(vad_parameters, dict):
                vad_parameters = VadOptions(**vad_parameters)
            speech_chunks = get_speech_timestamps(audio, vad_parameters, speech_probs=vad_speech_probs)
            audio_chunks, _ = collect_chunks(audio, speech_chunks)
            audio = np.concatenate(audio_chunks, axis=0)
            duration_after_vad = audio.shape[0] / sr
        features = self.feature_extractor(audio, chunk_length=chunk_length)

        all_language_probs = None
        if language is None:
            if not self.model.is_multilingual:
                language, language_probability = "en", 1
            else:
                first = float(clip_timestamps.split(",")[0]) if isinstance(clip_timestamps, str) else clip_timestamps[0]
                content_frames = features.shape[-1] - 1
                at = first * self.frames_per_second
                seek = int(at) if at < content_frames else 0
                language, language_probability, all_language_probs = self.detect_language(
                    features=features[..., seek:], language_detection_segments=language_detection_segments,
                    language_detection_threshold=language_detection_threshold)
        else:
            if not self.model.is_multilingual and language != "en":
                self.logger.warning("The current model is English-only but the language parameter is set to '%s'; "
                                    "using 'en' instead." % language)
                language = "en"
            language_probability = 1
        tokenizer = self.make_tokenizer(task=task, language=language)
        options = TranscriptionOptions(
            beam_size=beam_size, best_of=best_of, patience=patience, length_penalty=length_penalty,
            repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
            log_prob_threshold=log_prob_threshold, no_speech_threshold=no_speech_threshold,
            compression_ratio_threshold=compression_ratio_threshold,
            condition_on_previous_text=condition_on_previous_text,
            prompt_reset_on_temperature=prompt_reset_on_temperature,
            temperatures=(temperature if isinstance(temperature, (list, tuple)) else [temperature]),
            initial_prompt=initial_prompt, prefix=prefix, suppress_blank=suppress_blank,
            suppress_tokens=(get_suppressed_tokens(tokenizer, suppress_tokens) if suppress_tokens
                             else suppress_tokens),
            without_timestamps=without_timestamps, max_initial_timestamp=max_initial_timestamp,
            word_timestamps=word_timestamps, prepend_punctuations=prepend_punctuations,
            append_punctuations=append_punctuations, multilingual=multilingual, max_new_tokens=max_new_tokens,
            clip_timestamps=clip_timestamps, hallucination_silence_threshold=hallucination_silence_threshold,
            hotwords=hotwords)
        segments = self.generate_segments(features, tokenizer, options, log_progress, None)
        if speech_chunks:
            segments = restore_speech_timestamps(segments, speech_chunks, sr)
        info = TranscriptionInfo(language=language, language_probability=language_probability, duration=duration,
                                 duration_after_vad=duration_after_vad, transcription_options=options,
                                 vad_options=vad_parameters, all_language_probs=all_language_probs)
        return segments, info

    def generate_segments(self, features: np.ndarray, tokenizer: Tokenizer, options: TranscriptionOptions,
                          log_progress: bool = False, encoder_output: Optional[StorageView] = None):
        """The seek loop (generator of Segment).  For every clip [start, end) of `options.clip_timestamps`
        (frames; an odd count runs to the end of the audio) windows of up to 30 s are decoded at `seek`;
        the timestamps the model emitted (or the last word's end, or a hallucination-silence skip) decide
        where the next window starts."""
        from .words import get_end, is_segment_anomaly, next_words_segment
        fe = self.feature_extractor
        tpf, fps = fe.time_per_frame, self.frames_per_second
        content_frames = features.shape[-1] - 1
        content_duration = float(content_frames * tpf)
        if isinstance(options.clip_timestamps, str):
            options.clip_timestamps = [float(x) for x in options.clip_timestamps.split(",")] \
                if options.clip_timestamps else []
        marks = [round(t * fps) for t in options.clip_timestamps] or [0]
        if len(marks) % 2 == 1:
            marks.append(content_frames)
        clips = list(zip(marks[::2], marks[1::2]))

        all_tokens: List[int] = []
        prompt_reset_since = 0
        if options.initial_prompt is not None:
            if isinstance(options.initial_prompt, str):
                all_tokens.extend(tokenizer.encode(" " + options.initial_prompt.strip()))
            else:
                all_tokens.extend(options.initial_prompt)
        pbar = None
        if log_progress:
            from tqdm import tqdm
            pbar = tqdm(total=content_duration, unit="seconds")
        idx = 0
        last_speech_timestamp = 0.0
        for clip_start, clip_end in clips:
            clip_end = min(clip_end, content_frames)
            seek = clip_start
            while True:
                seek = max(seek, clip_start)
                if seek >= clip_end:
                    break
                previous_seek = seek
                time_offset = seek * tpf
                window_end_time = float((seek + fe.nb_max_frames) * tpf)
                segment_size = min(fe.nb_max_frames, content_frames - seek, clip_end - seek)
                segment_duration = segment_size * tpf
                window = pad_or_trim(features[:, seek:seek + segment_size])
                previous_tokens = all_tokens[prompt_reset_since:]
                if seek > 0 or encoder_output is None:
                    encoder_output = self.encode(window)
                if options.multilingual:
                    language_token, _ = self.model.detect_language(encoder_output)[0][0]
                    names = language_token_strings(self.model.config)
                    tokenizer.language = self.model.config.lang_begin + names.index(language_token)
                    tokenizer.language_code = language_token[2:-2]
                prompt = self.get_prompt(tokenizer, previous_tokens, without_timestamps=options.without_timestamps,
                                         prefix=options.prefix if seek == 0 else None, hotwords=options.hotwords)
                result, avg_logprob, temperature, compression_ratio = self.generate_with_fallback(
                    encoder_output, prompt, tokenizer, options)

                if options.no_speech_threshold is not None:
                    skip = result.no_speech_prob > options.no_speech_threshold
                    if options.log_prob_threshold is not None and avg_logprob > options.log_prob_threshold:
                        skip = False            # confident text beats the no-speech probability
                    if skip:
                        seek += segment_size    # fast-forward to the next window
                        continue

                tokens = result.sequences_ids[0]
                current, seek, single_timestamp_ending = self._split_segments_by_timestamps(
                    tokenizer=tokenizer, tokens=tokens, time_offset=time_offset, segment_size=segment_size,
                    segment_duration=segment_duration, seek=seek)

                if options.word_timestamps:
                    self.add_word_timestamps([current], tokenizer, encoder_output, segment_size,
                                             options.prepend_punctuations, options.append_punctuations,
                                             last_speech_timestamp=last_speech_timestamp)
                    if not single_timestamp_ending:
                        last_word_end = get_end(current)
                        if last_word_end is not None and last_word_end > time_offset:
                            seek = round(last_word_end * fps)
                    threshold = options.hallucination_silence_threshold
                    if threshold is not None:
                        # leading silence before a probable hallucination: re-decode from where speech starts
                        first = next_words_segment(current)
                        if first is not None and is_segment_anomaly(first):
                            gap = first["start"] - time_offset
                            if gap > threshold:
                                seek = previous_seek + round(gap * fps)
                                continue
                        # a probable hallucination surrounded by silence (or more of them): cut there
                        hal_last_end = last_speech_timestamp
                        for si, seg in enumerate(current):
                            if not seg["words"]:
                                continue
                            if is_segment_anomaly(seg):
                                nxt = next_words_segment(current[si + 1:])
                                hal_next_start = nxt["words"][0]["start"] if nxt is not None \
                                    else time_offset + segment_duration
                                silence_before = (seg["start"] - hal_last_end > threshold or seg["start"] < threshold
                                                  or seg["start"] - time_offset < 2.0)
                                silence_after = (hal_next_start - seg["end"] > threshold or is_segment_anomaly(nxt)
                                                 or window_end_time - seg["end"] < 2.0)
                                if silence_before and silence_after:
                                    seek = round(max(time_offset + 1, seg["start"]) * fps)
                                    if content_duration - seg["end"] < threshold:
                                        seek = content_frames
                                    current[si:] = []
                                    break
                            hal_last_end = seg["end"]
                    last_word_end = get_end(current)
                    if last_word_end is not None:
                        last_speech_timestamp = last_word_end

                for seg in current:
                    text = tokenizer.decode(seg["tokens"])
                    if seg["start"] == seg["end"] or not text.strip():
                        continue
                    all_tokens.extend(seg["tokens"])
                    idx += 1
                    yield Segment(id=idx, seek=previous_seek, start=seg["start"], end=seg["end"], text=text,
                                  tokens=seg["tokens"], temperature=temperature, avg_logprob=avg_logprob,
                                  compression_ratio=compression_ratio, no_speech_prob=result.no_speech_prob,
                                  words=([Word(**w) for w in seg["words"]] if options.word_timestamps else None))

                if not options.condition_on_previous_text or temperature > options.prompt_reset_on_temperature:
                    prompt_reset_since = len(all_tokens)
                if pbar is not None:
                    pbar.update((min(content_frames, seek) - previous_seek) * tpf)
        if pbar is not None:
            pbar.close()

    def generate_with_fallback(self, encoder_output: StorageView, prompt: List[int], tokenizer: Tokenizer,
                               options: TranscriptionOptions):
        """Temperature ladder (transcribe.py:1402-1530): beam search at t = 0, `best_of` random samples at
        t > 0; a result is accepted unless it is too repetitive (compression ratio) or too improbable
        (average log-prob) — except when it looks like silence.  If every temperature fails, the most probable
        attempt (preferring those under the compression threshold) is returned with the last temperature.
        -> (WhisperGenerationResult, avg_logprob, temperature, compression_ratio)"""
        max_initial_timestamp_index = int(round(options.max_initial_timestamp / self.time_precision))
        max_length = len(prompt) + options.max_new_tokens if options.max_new_tokens is not None else self.max_length
        if max_length > self.max_length:
            raise ValueError(
                f"The length of the prompt is {len(prompt)}, and the `max_new_tokens` {max_length - len(prompt)}. "
                f"Thus, the combined length of the prompt and `max_new_tokens` is: {max_length}. This exceeds the "
                f"`max_length` of the Whisper model: {self.max_length}. You should either reduce the length of your "
                f"prompt, or reduce the value of `max_new_tokens`, so that their combined length is less that "
                f"{self.max_length}.")
        attempts, compact = [], []
        chosen = None
        for temperature in options.temperatures:
            if temperature > 0:
                mode = dict(beam_size=1, num_hypotheses=options.best_of, sampling_topk=0,
                            sampling_temperature=temperature)
            else:
                mode = dict(beam_size=options.beam_size, patience=options.patience)
            result = self.model.generate(
                encoder_output, [prompt], length_penalty=options.length_penalty,
                repetition_penalty=options.repetition_penalty, no_repeat_ngram_size=options.no_repeat_ngram_size,
                max_length=max_length, return_scores=True, return_no_speech_prob=True,
                suppress_blank=options.suppress_blank, suppress_tokens=options.suppress_tokens,
                max_initial_timestamp_index=max_initial_timestamp_index, **mode)[0]
            tokens = result.sequences_ids[0]
            n = len(tokens)
            avg_logprob = result.scores[0] * (n ** options.length_penalty) / (n + 1)   # undo the length norm
            compression_ratio = get_compression_ratio(tokenizer.decode(tokens).strip())
            attempt = (result, avg_logprob, temperature, compression_ratio)
            attempts.append(attempt)
            retry = False
            if options.compression_ratio_threshold is not None:
                if compression_ratio > options.compression_ratio_threshold:
                    retry = True
                else:
                    compact.append(attempt)
            low_prob = options.log_prob_threshold is not None and avg_logprob < options.log_prob_threshold
            if low_prob:
                retry = True
            if options.no_speech_threshold is not None and result.no_speech_prob > options.no_speech_threshold \
                    and low_prob:
                retry = False                   # silence: a colder decode will not help
            if not retry:
                chosen = attempt
                break
        if chosen is None:
            best = max(compact or attempts, key=lambda a: a[1])
            chosen = (best[0], best[1], temperature, best[3])   # last temperature drives the prompt reset
        return chosen

    def make_tokenizer(self, task="transcribe", language="en") -> Tokenizer:
        return Tokenizer(self.hf_tokenizer, self.model.config, self.model.is_multilingual, task=task, language=language)

    # ---- backend wrappers -----------------------------------------------------------------
    def encode(self, features: np.ndarray) -> StorageView:
        to_cpu = self.model.device == "cuda" and len(self.model.device_index) > 1
        if features.ndim == 2:
            features = np.expand_dims(features, 0)
        return self.model.encode(StorageView.from_array(np.ascontiguousarray(features)), to_cpu=to_cpu)

    def get_prompt(self, tokenizer: Tokenizer, previous_tokens: List[int], without_timestamps: bool = False,
                   prefix: Optional[str] = None, hotwords: Optional[str] = None) -> List[int]:
        prompt = []
        half = self.max_length // 2
        if previous_tokens or (hotwords and not prefix):
            prompt.append(tokenizer.sot_prev)
            if hotwords and not prefix:
                hw = tokenizer.encode(" " + hotwords.strip())
                prompt.extend(hw[:half - 1] if len(hw) >= half else hw)
            if previous_tokens:
                prompt.extend(previous_tokens[-(half - 1):])
        prompt.extend(tokenizer.sot_sequence)
        if without_timestamps:
            prompt.append(tokenizer.no_timestamps)
        if prefix:
            pt = tokenizer.encode(" " + prefix.strip())
            if len(pt) >= half:
                pt = pt[:half - 1]
            if not without_timestamps:
                prompt.append(tokenizer.timestamp_begin)
            prompt.extend(pt)
        return prompt

    def _split_segments_by_timestamps(self, tokenizer: Tokenizer, tokens: List[int], time_offset: float,
                                      segment_size: int, segment_duration: float, seek: int):
        """consecutive-timestamp slicing of one chunk's tokens (transcribe.py:1024-1101)"""
        tb = tokenizer.timestamp_begin
        segs = []
        single_ts_end = len(tokens) >= 2 and tokens[-2] < tb <= tokens[-1]
        cuts = [i for i in range(1, len(tokens)) if tokens[i] >= tb and tokens[i - 1] >= tb]
        if cuts:
            if single_ts_end:
                cuts.append(len(tokens))
            last = 0
            for cur in cuts:
                piece = tokens[last:cur]
                segs.append(dict(seek=seek, start=time_offset + (piece[0] - tb) * self.time_precision,
                                 end=time_offset + (piece[-1] - tb) * self.time_precision, tokens=piece))
                last = cur
            if single_ts_end:
                seek += segment_size
            else:
                seek += (tokens[last - 1] - tb) * self.input_stride
        else:
            duration = segment_duration
            stamps = [t for t in tokens if t >= tb]
            if stamps and stamps[-1] != tb:
                duration = (stamps[-1] - tb) * self.time_precision
            segs.append(dict(seek=seek, start=time_offset, end=time_offset + duration, tokens=tokens))
            seek += segment_size
        return segs, seek, single_ts_end

    def find_alignment(self, tokenizer: Tokenizer, text_tokens: List[List[int]], encoder_output: StorageView,
                       num_frames, median_filter_width: int = 7) -> List[dict]:
        """per chunk: the words {word, tokens, start, end, probability} of its text tokens, timed by the
        backend's cross-attention DTW (transcribe.py:1698-1766).  `num_frames`: int or one int per chunk."""
        from .words import word_alignment
        if len(text_tokens) == 0:
            return []
        results = self.model.align(encoder_output, tokenizer.sot_sequence, text_tokens, num_frames,
                                   median_filter_width=median_filter_width)
        return [word_alignment(tokenizer, toks, res.alignments, res.text_token_probs, self.tokens_per_second)
                for res, toks in zip(results, text_tokens)]

    def add_word_timestamps(self, segments: List[List[dict]], tokenizer: Tokenizer, encoder_output: StorageView,
                            num_frames, prepend_punctuations: str, append_punctuations: str,
                            last_speech_timestamp: float) -> float:
        """`segments`: per chunk, the list of its sub-segment dicts (seek/start/end/tokens).  Aligns every
        chunk's text tokens in ONE backend call, merges punctuation into neighbouring words, clamps
        over-long words at sentence / pause / segment boundaries and stores `words` in every sub-segment
        (transcribe.py:1567-1696).  -> end time of the last word (carried into the next call)."""
        if len(segments) == 0:
            return None
        aligned = self.align_words(segments, tokenizer, encoder_output, num_frames, prepend_punctuations,
                                   append_punctuations)
        return self.apply_word_alignments(segments, aligned, last_speech_timestamp)

    def align_words(self, segments: List[List[dict]], tokenizer: Tokenizer, encoder_output: StorageView, num_frames,
                    prepend_punctuations: str, append_punctuations: str) -> List[dict]:
        """Chunk-local half of add_word_timestamps: one backend `align` call for all chunks, then per chunk the
        word list with sentence-boundary clamping and punctuation merged.  Needs the encoder output, i.e. runs
        on the rank that decoded the chunks; the result is plain data (picklable) for `apply_word_alignments`."""
        from .words import clamp_sentence_boundaries, merge_punctuations
        per_sub = [[[t for t in sub["tokens"] if t < tokenizer.eot] for sub in chunk] for chunk in segments]
        text_tokens = [[t for sub in subs for t in sub] for subs in per_sub]
        alignments = self.find_alignment(tokenizer, text_tokens, encoder_output, num_frames)
        out = []
        for alignment, subs in zip(alignments, per_sub):
            median, longest = clamp_sentence_boundaries(alignment)
            merge_punctuations(alignment, prepend_punctuations, append_punctuations)
            out.append(dict(words=alignment, median=median, longest=longest, tokens_per_subsegment=subs))
        return out

    def apply_word_alignments(self, segments: List[List[dict]], aligned: List[dict],
                              last_speech_timestamp: float) -> float:
        """Sequential half: distributes every chunk's words over its sub-segments; the pause heuristics chain
        through `last_speech_timestamp` from chunk to chunk, so this runs in chunk order on one process."""
        from .words import assign_words
        for chunk, a in zip(segments, aligned):
            last_speech_timestamp = assign_words(chunk, a["words"], a["tokens_per_subsegment"],
                                                 chunk[0]["seek"] / self.frames_per_second, a["median"], a["longest"],
                                                 last_speech_timestamp)
        return last_speech_timestamp

    def detect_language(self, audio: Optional[np.ndarray] = None, features: Optional[np.ndarray] = None,
                        vad_filter: bool = False, vad_parameters=None, language_detection_segments: int = 1,
                        language_detection_threshold: float = 0.5):
        assert audio is not None or features is not None, "Either `audio` or `features` must be provided."
        fe = self.feature_extractor
        if audio is not None:
            if vad_filter:   # keep only the speech before looking at the first segments (transcribe.py:1802-1806)
                from .vad import VadOptions, collect_chunks, get_speech_timestamps
                if isinstance(vad_parameters, dict):
                    vad_parameters = VadOptions(**vad_parameters)
                speech_chunks = get_speech_timestamps(audio, vad_parameters)
                audio = np.concatenate(collect_chunks(audio, speech_chunks)[0], axis=0)
            features = fe(audio[: language_detection_segments * fe.n_samples])
        features = features[..., : language_detection_segments * fe.nb_max_frames]
        seen = {}
        for i in range(0, features.shape[-1], fe.nb_max_frames):
            enc = self.encode(pad_or_trim(features[..., i:i + fe.nb_max_frames]))
            results = self.model.detect_language(enc)[0]
            all_probs = [(tok[2:-2], p) for tok, p in results]
            language, prob = all_probs[0]
            if prob > language_detection_threshold:
                break
            seen.setdefault(language, []).append(prob)
        else:
            language = max(seen, key=lambda k: len(seen[k]))
            prob = max(seen[language])
        return language, prob, all_probs


class BatchedInferencePipeline:
    def __init__(self, model: WhisperModel):
        self.model = model
        self.last_speech_timestamp = 0.0

    # ---- one batch: encode + generate ---------------------------------------------------
    def generate_segment_batched(self, features, tokenizer: Tokenizer, options: TranscriptionOptions,
                                 audio_chunks: Optional[List[np.ndarray]] = None):
        """features: [B, n_mels, 3000] float32, or None when `audio_chunks` is given (fused resident
        PCM -> log-mel -> encoder path; the features never leave HBM)."""
        m = self.model
        batch_size = len(audio_chunks) if audio_chunks is not None else features.shape[0]
        prompt = m.get_prompt(tokenizer,
                              previous_tokens=(tokenizer.encode(options.initial_prompt)
                                               if isinstance(options.initial_prompt, str)
                                               else list(options.initial_prompt or [])),
                              without_timestamps=options.without_timestamps, hotwords=options.hotwords)
        max_length = len(prompt) + options.max_new_tokens if options.max_new_tokens is not None else m.max_length
        if max_length > m.max_length:
            raise ValueError(
                f"The length of the prompt is {len(prompt)}, and the `max_new_tokens` {max_length - len(prompt)}. "
                f"Thus, the combined length of the prompt and `max_new_tokens` is: {max_length}. This exceeds the "
                f"`max_length` of the Whisper model: {m.max_length}. You should either reduce the length of your "
                f"prompt, or reduce the value of `max_new_tokens`, so that their combined length is less that "
                f"{m.max_length}.")
        encoder_output = m.model.encode_pcm(audio_chunks) if audio_chunks is not None else m.encode(features)
        prompts = [prompt.copy() for _ in range(batch_size)]
        if options.multilingual:
            names = language_token_strings(m.model.config)
            lang_tokens = [m.model.config.lang_begin + names.index(r[0][0]) for r in
                           m.model.detect_language(encoder_output)]
            idx = prompt.index(tokenizer.language)
            for i, t in enumerate(lang_tokens):
                prompts[i][idx] = t
        results = m.model.generate(
            encoder_output, prompts, beam_size=options.beam_size, patience=options.patience,
            length_penalty=options.length_penalty, max_length=max_length, suppress_blank=options.suppress_blank,
            suppress_tokens=options.suppress_tokens, return_scores=True, return_no_speech_prob=True,
            sampling_temperature=options.temperatures[0], repetition_penalty=options.repetition_penalty,
            no_repeat_ngram_size=options.no_repeat_ngram_size)
        output = []
        for r in results:
            n = len(r.sequences_ids[0])
            cum = r.scores[0] * (n ** options.length_penalty)
            output.append(dict(avg_logprob=cum / (n + 1), no_speech_prob=r.no_speech_prob, tokens=r.sequences_ids[0]))
        return encoder_output, output

    def forward(self, features, tokenizer, chunks_metadata, options, audio_chunks=None):
        encoder_output, outputs = self.generate_segment_batched(features, tokenizer, options, audio_chunks)
        return self._segment_outputs(outputs, tokenizer, chunks_metadata, options, encoder_output)

    def _segment_outputs(self, outputs, tokenizer, chunks_metadata, options, encoder_output=None):
        m = self.model
        segmented, sizes = self._split_outputs(outputs, tokenizer, chunks_metadata)
        if options.word_timestamps:
            self.last_speech_timestamp = m.add_word_timestamps(
                segmented, tokenizer, encoder_output, sizes, options.prepend_punctuations,
                options.append_punctuations, self.last_speech_timestamp)
        return segmented

    def _split_outputs(self, outputs, tokenizer, chunks_metadata):
        """per chunk: its sub-segment dicts (timestamp splitting) and its size in frames"""
        m = self.model
        segmented, sizes = [], []
        for meta, out in zip(chunks_metadata, outputs):
            duration = meta["duration"]
            size = int(ceil(duration) * m.frames_per_second)
            sizes.append(size)
            subs, _, _ = m._split_segments_by_timestamps(tokenizer=tokenizer, tokens=out["tokens"],
                                                         time_offset=meta["offset"], segment_size=size,
                                                         segment_duration=duration, seek=0)
            segmented.append([
                dict(text=tokenizer.decode(s["tokens"]), avg_logprob=out["avg_logprob"],
                     no_speech_prob=out["no_speech_prob"], tokens=s["tokens"], start=s["start"], end=s["end"],
                     compression_ratio=get_compression_ratio(tokenizer.decode(s["tokens"])),
                     seek=int(meta["offset"] * m.frames_per_second))
                for s in subs])
        return segmented, sizes

    # ---- the public entry point ---------------------------------------------------------
    def transcribe(self, audio: Union[str, np.ndarray], language: Optional[str] = None, task: str = "transcribe",
                   log_progress: bool = False, beam_size: int = 5, best_of: int = 5, patience: float = 1,
                   length_penalty: float = 1, repetition_penalty: float = 1, no_repeat_ngram_size: int = 0,
                   temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), compression_ratio_threshold: Optional[float] = 2.4,
                   log_prob_threshold: Optional[float] = -1.0, no_speech_threshold: Optional[float] = 0.6,
                   condition_on_previous_text: bool = True, prompt_reset_on_temperature: float = 0.5,
                   initial_prompt=None, prefix: Optional[str] = None, suppress_blank: bool = True,
                   suppress_tokens: Optional[List[int]] = [-1], without_timestamps: bool = True,
                   max_initial_timestamp: float = 1.0, word_timestamps: bool = False,
                   prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
                   multilingual: bool = False, vad_filter: bool = True, vad_parameters=None,
                   max_new_tokens: Optional[int] = None, chunk_length: Optional[int] = None,
                   clip_timestamps: Optional[List[dict]] = None, hallucination_silence_threshold=None,
                   batch_size: int = 8, hotwords: Optional[str] = None,
                   language_detection_threshold: Optional[float] = 0.5, language_detection_segments: int = 1,
                   shard: bool = False, fused_features: bool = True, vad_speech_probs=None):
        """Same contract as the reference (transcribe.py:254-578): returns (segment generator, info).
        shard=True: inside a torch.distributed job every rank calls this with the same arguments; the
        chunk list is block-partitioned over the ranks and rank 0's generator yields ALL segments in
        order (other ranks yield nothing).  fused_features=False reproduces the reference data flow
        (features materialised on the host, then encode())."""
        m = self.model
        sr = m.feature_extractor.sampling_rate
        if multilingual and not m.model.is_multilingual:
            m.logger.warning("The current model is English-only but the multilingual parameter is set to"
                             "True; setting to False instead.")
            multilingual = False
        if not isinstance(audio, np.ndarray):
            from .audio import decode_audio   # path / file object: WAVE and FLAC natively, other containers through PyAV
            audio = decode_audio(audio, sampling_rate=sr)
        audio = np.asarray(audio, dtype=np.float32)
        duration = audio.shape[0] / sr
        chunk_length = chunk_length or m.feature_extractor.chunk_length
        from .vad import VadOptions, collect_chunks, get_speech_timestamps
        from .words import restore_speech_timestamps
        clips_given = bool(clip_timestamps)
        if not clips_given:
            # no split provided: speech spans from the VAD (merged into <= chunk_length chunks), or the whole
            # audio when it is shorter than one chunk
            if vad_filter:
                if vad_parameters is None:
                    vad_parameters = VadOptions(max_speech_duration_s=chunk_length, min_silence_duration_ms=160)
                elif isinstance(vad_parameters, dict):
                    vad_parameters = VadOptions(**{k: v for k, v in vad_parameters.items()
                                                   if k != "max_speech_duration_s"},
                                                max_speech_duration_s=chunk_length)
                clips = get_speech_timestamps(audio, vad_parameters, speech_probs=vad_speech_probs)
            elif duration < chunk_length:
                clips = [{"start": 0, "end": audio.shape[0]}]
            else:
                raise RuntimeError("No clip timestamps found. Set 'vad_filter' to True or provide 'clip_timestamps'.")
            audio_chunks, chunks_metadata = collect_chunks(audio, clips, max_duration=chunk_length)
        else:
            clips = [{k: int(v * sr) for k, v in seg.items()} for seg in clip_timestamps]
            audio_chunks, chunks_metadata = [], []
            for i, clip in enumerate(clips):
                audio_chunks.append(audio[clip["start"]:clip["end"]])
                d = (clip["end"] - clip["start"]) / sr
                if d > 30:
                    m.logger.warning("Segment %d is longer than 30 seconds, only the first 30 seconds will be "
                                     "transcribed", i)
                chunks_metadata.append({"offset": clip["start"] / sr, "duration": d, "segments": [clip]})
        duration_after_vad = sum(c["end"] - c["start"] for c in clips) / sr
        # the reference formats the removed duration for its log line, which asserts it is not negative
        # (transcribe.py:458-461, utils.py:124): clips that add up to more than the recording are rejected
        assert duration - duration_after_vad >= 0, "non-negative timestamp expected"
        if not duration_after_vad:
            audio_chunks, chunks_metadata = [], []

        all_language_probs = None
        if language is None:
            if not m.model.is_multilingual:
                language, language_probability = "en", 1
            else:
                # the reference concatenates the (unpadded) features of ALL chunks; detect_language only looks
                # at the first language_detection_segments * 3000 frames, so stop once those are covered
                need = language_detection_segments * m.feature_extractor.nb_max_frames
                parts, have = [], 0
                for chunk in audio_chunks:
                    if have >= need:
                        break
                    parts.append(m.feature_extractor(chunk)[..., :-1])
                    have += parts[-1].shape[-1]
                # + one dummy frame so that empty audio still has a feature
                feats = np.concatenate(parts + [np.full((m.model.n_mels, 1), -1.5, dtype="float32")], axis=1)
                language, language_probability, all_language_probs = m.detect_language(
                    features=feats, language_detection_segments=language_detection_segments,
                    language_detection_threshold=language_detection_threshold)
        else:
            if not m.model.is_multilingual and language != "en":
                m.logger.warning("The current model is English-only but the language parameter is set to '%s'; "
                                 "using 'en' instead." % language)
                language = "en"
            language_probability = 1
        tokenizer = m.make_tokenizer(task=task, language=language)
        options = TranscriptionOptions(
            beam_size=beam_size, best_of=best_of, patience=patience, length_penalty=length_penalty,
            repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
            log_prob_threshold=log_prob_threshold, no_speech_threshold=no_speech_threshold,
            compression_ratio_threshold=compression_ratio_threshold,
            temperatures=(list(temperature[:1]) if isinstance(temperature, (list, tuple)) else [temperature]),
            initial_prompt=initial_prompt, prefix=prefix, suppress_blank=suppress_blank,
            suppress_tokens=(get_suppressed_tokens(tokenizer, list(suppress_tokens)) if suppress_tokens
                             else suppress_tokens),
            prepend_punctuations=prepend_punctuations, append_punctuations=append_punctuations,
            max_new_tokens=max_new_tokens, hotwords=hotwords, word_timestamps=word_timestamps,
            hallucination_silence_threshold=None, condition_on_previous_text=False,
            clip_timestamps=(clip_timestamps if clips_given else clips), prompt_reset_on_temperature=0.5,
            multilingual=multilingual,
            without_timestamps=without_timestamps, max_initial_timestamp=0.0)
        info = TranscriptionInfo(language=language, language_probability=language_probability, duration=duration,
                                 duration_after_vad=duration_after_vad, transcription_options=options,
                                 vad_options=vad_parameters, all_language_probs=all_language_probs)
        gen = self._batched_segments_generator(audio_chunks, tokenizer, chunks_metadata, batch_size, options,
                                               log_progress, shard, fused_features)
        if not clips_given:
            gen = restore_speech_timestamps(gen, clips, sr)
        return gen, info

    def _batched_segments_generator(self, audio_chunks, tokenizer, chunks_metadata, batch_size, options,
                                    log_progress, shard=False, fused_features=True):
        from .sharding import gather_results, partition
        m = self.model
        rank, world, local_rank = 0, 1, 0
        if shard:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        n = len(audio_chunks)
        seg_idx = 0
        # The batches of a recording are independent (`condition_on_previous_text=False`): with worker replicas
        # (WhisperModel(num_workers=W) -> backend inter_threads) W batches are kept in flight on the GPU — their
        # encoders run side by side and their generate() calls share decode runs (backend decode groups).  Only
        # the word-timestamp pause heuristics chain through last_speech_timestamp; those run in order, below.
        workers = int(getattr(m.model, "inter_threads", 1) or 1)

        def decode_batch(i0, i1):
            chunks = audio_chunks[i0:i1]
            feats = None if fused_features else m.model.log_mel(chunks)
            enc, outs = self.generate_segment_batched(feats, tokenizer, options,
                                                      audio_chunks=chunks if fused_features else None)
            local = aligned = None
            if not shard or options.word_timestamps:
                local, sizes = self._split_outputs(outs, tokenizer, chunks_metadata[i0:i1])
                if options.word_timestamps:
                    # the chunk-local half of the word timing runs where the encoder output lives
                    aligned = m.align_words(local, tokenizer, enc, sizes, options.prepend_punctuations,
                                            options.append_punctuations)
            return outs, local, aligned

        def batches(lo, hi):
            """(outs, local, aligned) per batch of the chunk range [lo, hi), in order, `workers` batches in flight"""
            spans = [(i, min(hi, i + batch_size)) for i in range(lo, hi, batch_size)]
            if workers >= 2 and len(spans) == 1 and hi - lo >= 4:
                # ONE batch and idle workers (a rank's share of a sharded recording: 1 h over 8 GPUs = 15 chunks): two
                # half batches on two workers — the second half's encoder pass runs under the first half's decode run and
                # the two runs decode side by side on the group's two lanes.  Results do not depend on the split (every
                # kernel works per chunk); measured 315 -> 290 ms for 15 chunks (profiles/r06_bench_c4_halves.json)
                mid = lo + (hi - lo + 1) // 2
                spans = [(lo, mid), (mid, hi)]
            if workers <= 1 or len(spans) <= 1:
                for sp in spans:
                    yield decode_batch(*sp)
                return
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=workers) as pool:
                pending = deque()
                for sp in spans:
                    pending.append(pool.submit(decode_batch, *sp))
                    if len(pending) >= workers:
                        yield pending.popleft().result()
                while pending:
                    yield pending.popleft().result()

        def emit(results):
            nonlocal seg_idx
            for result in results:
                for seg in result:
                    seg_idx += 1
                    yield Segment(seek=seg["seek"], id=seg_idx, text=seg["text"], start=round(seg["start"], 3),
                                  end=round(seg["end"], 3), tokens=seg["tokens"], avg_logprob=seg["avg_logprob"],
                                  words=(None if not options.word_timestamps else [Word(**w) for w in seg["words"]]),
                                  no_speech_prob=seg["no_speech_prob"], compression_ratio=seg["compression_ratio"],
                                  temperature=options.temperatures[0])

        if not shard:
            # single process: segments are yielded as soon as their batch is decoded
            for _, results, aligned in batches(0, n):
                if options.word_timestamps:
                    self.last_speech_timestamp = m.apply_word_alignments(results, aligned, self.last_speech_timestamp)
                yield from emit(results)
            self.last_speech_timestamp = 0.0
            return
        # sharded (a job of ONE rank takes this path too: same result, and the way a 1-GPU box executes the RCCL
        # gather): every rank decodes its contiguous block of the chunk list (no collective in the data path), then
        # ONE gather brings the fixed-size result records — and the chunk-local word alignments — to rank 0, in
        # rank order = the serial order
        bounds = partition(n, world)
        lo, hi = bounds[rank]
        max_len = m.max_length
        my_outs, my_aligned = [], []
        for outs, _, aligned in batches(lo, hi):
            my_outs.extend(outs)
            if options.word_timestamps:
                my_aligned.extend(aligned)
        counts = [b[1] - b[0] for b in bounds]
        ordered_aligned = None
        if options.word_timestamps:
            import torch.distributed as dist
            bucket = [None] * world if rank == 0 else None
            dist.gather_object(my_aligned, bucket, dst=0)
            if rank == 0:
                ordered_aligned = [a for r in range(world) for a in bucket[r]]
        got = gather_results(_OutRec.wrap(my_outs), max_len, rank, world, local_rank, counts=counts)
        if rank != 0:
            return
        ordered = [dict(tokens=ids, avg_logprob=avg_lp, no_speech_prob=nsp) for (ids, avg_lp, nsp) in got]
        results, _ = self._split_outputs(ordered, tokenizer, chunks_metadata)
        if options.word_timestamps:
            self.last_speech_timestamp = m.apply_word_alignments(results, ordered_aligned, self.last_speech_timestamp)
        yield from emit(results)
        self.last_speech_timestamp = 0.0


class _OutRec:
    """adapts the pipeline's per-chunk dicts to the record layout of sharding.gather_results
    (ids, one float score slot = avg_logprob, no_speech_prob)"""

    def __init__(self, d):
        self.sequences_ids = [d["tokens"]]
        self.scores = [d["avg_logprob"]]
        self.no_speech_prob = d["no_speech_prob"]

    @staticmethod
    def wrap(outs):
        return [_OutRec(o) for o in outs]
