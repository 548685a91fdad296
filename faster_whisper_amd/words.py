"""Word-level timing on top of the backend's `align` and the timestamp
restoration after VAD chunking (row a13).  Host-side float/integer logic, same behaviour as the reference:

    word_alignment            transcribe.py:1717-1766 (the body of find_alignment after the backend call)
    merge_punctuations        transcribe.py:1910-1943
    clamp_sentence_boundaries transcribe.py:1594-1618 (median / max word duration, sentence-end truncation)
    assign_words              transcribe.py:1620-1696 (words -> sub-segments, pause heuristics, segment bounds)
    word_anomaly_score / is_segment_anomaly / next_words_segment
                              transcribe.py:1226-1250 (hallucination heuristics of the sequential path)
    get_end                   utils.py:148-152
    restore_speech_timestamps transcribe.py:1843-1870
"""
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .vad import SpeechTimestampsMap

SENTENCE_END = ".。!！?？"
PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"


def word_anomaly_score(word: dict) -> float:
    """improbable (< 0.15), very short (< 133 ms) or very long (> 2 s) words look hallucinated"""
    duration = word["end"] - word["start"]
    score = 1.0 if word.get("probability", 0.0) < 0.15 else 0.0
    if duration < 0.133:
        score += (0.133 - duration) * 15
    if duration > 2.0:
        score += duration - 2.0
    return score


def is_segment_anomaly(segment: Optional[dict]) -> bool:
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in PUNCTUATION][:8]
    score = sum(word_anomaly_score(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def next_words_segment(segments: List[dict]) -> Optional[dict]:
    return next((s for s in segments if s["words"]), None)


def word_alignment(tokenizer, text_tokens: List[int], alignments: Sequence[Tuple[int, int]],
                   text_token_probs: Sequence[float], tokens_per_second: int) -> List[dict]:
    """One chunk: DTW path (text index, frame index) + token probabilities -> list of words
    {word, tokens, start, end, probability} with times in seconds from the chunk start."""
    words, word_tokens = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(word_tokens) <= 1:           # only <|endoftext|>
        return []
    bounds = np.concatenate([[0], np.cumsum([len(t) for t in word_tokens[:-1]])]).astype(np.int64)
    if len(bounds) <= 1:
        return []
    text_idx = np.array([p[0] for p in alignments])
    time_idx = np.array([p[1] for p in alignments])
    # the first path point of every text token (= where the path moves to the next token)
    first = np.concatenate([[True], np.diff(text_idx).astype(bool)])
    token_time = time_idx[first] / tokens_per_second
    starts, ends = token_time[bounds[:-1]], token_time[bounds[1:]]
    probs = [np.mean(text_token_probs[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    return [dict(word=w, tokens=t, start=s, end=e, probability=p)
            for w, t, s, e, p in zip(words, word_tokens, starts, ends, probs)]


def merge_punctuations(alignment: List[dict], prepended: str, appended: str) -> None:
    """In place: glue opening punctuation to the following word (right-to-left pass) and closing punctuation
    to the preceding word (left-to-right pass); absorbed entries keep their slot with word "" / tokens []."""
    nxt = len(alignment) - 1
    for cur in range(len(alignment) - 2, -1, -1):
        a, b = alignment[cur], alignment[nxt]
        if a["word"].startswith(" ") and a["word"].strip() in prepended:
            b["word"] = a["word"] + b["word"]
            b["tokens"] = a["tokens"] + b["tokens"]
            a["word"], a["tokens"] = "", []
        else:
            nxt = cur
    prv = 0
    for cur in range(1, len(alignment)):
        a, b = alignment[prv], alignment[cur]
        if not a["word"].endswith(" ") and b["word"] in appended:
            a["word"] = a["word"] + b["word"]
            a["tokens"] = a["tokens"] + b["tokens"]
            b["word"], b["tokens"] = "", []
        else:
            prv = cur


def clamp_sentence_boundaries(alignment: List[dict]) -> Tuple[float, float]:
    """-> (median_duration capped at 0.7 s, max_duration = 2 x median); words at a sentence boundary that
    are longer than max_duration are truncated (in place)."""
    durations = np.array([w["end"] - w["start"] for w in alignment])
    durations = durations[durations.nonzero()]
    median = min(0.7, float(np.median(durations) if len(durations) > 0 else 0.0))
    longest = median * 2
    if len(durations) > 0:
        for i in range(1, len(alignment)):
            w = alignment[i]
            if w["end"] - w["start"] > longest:
                if w["word"] in SENTENCE_END:
                    w["end"] = w["start"] + longest
                elif alignment[i - 1]["word"] in SENTENCE_END:
                    w["start"] = w["end"] - longest
    return median, longest


def assign_words(subsegments: List[dict], alignment: List[dict], tokens_per_subsegment: List[List[int]],
                 time_offset: float, median: float, longest: float, last_speech_timestamp: float) -> float:
    """Distributes one chunk's aligned words over its sub-segments (by token count), applies the
    after-a-pause and segment-boundary heuristics and writes `words` / adjusted `start` / `end` into the
    sub-segment dicts.  -> updated last_speech_timestamp."""
    wi = 0
    for sub, sub_tokens in zip(subsegments, tokens_per_subsegment):
        taken = 0
        words = []
        while wi < len(alignment) and taken < len(sub_tokens):
            t = alignment[wi]
            if t["word"]:
                words.append(dict(word=t["word"], start=round(time_offset + t["start"], 2),
                                  end=round(time_offset + t["end"], 2), probability=t["probability"]))
            taken += len(t["tokens"])
            wi += 1
        if words:
            first = words[0]
            # the first (and second) word after a long pause must not be longer than twice the median
            after_pause = first["end"] - last_speech_timestamp > median * 4
            too_long = first["end"] - first["start"] > longest or (
                len(words) > 1 and words[1]["end"] - first["start"] > longest * 2)
            if after_pause and too_long:
                if len(words) > 1 and words[1]["end"] - words[1]["start"] > longest:
                    boundary = max(words[1]["end"] / 2, words[1]["end"] - longest)
                    first["end"] = words[1]["start"] = boundary
                first["start"] = max(0, first["end"] - longest)
            # prefer the segment-level start if the first word is too long
            if sub["start"] < first["end"] and sub["start"] - 0.5 > first["start"]:
                first["start"] = max(0, min(first["end"] - median, sub["start"]))
            else:
                sub["start"] = first["start"]
            last = words[-1]
            # prefer the segment-level end if the last word is too long
            if sub["end"] > last["start"] and sub["end"] + 0.5 < last["end"]:
                last["end"] = max(last["start"] + median, sub["end"])
            else:
                sub["end"] = last["end"]
            last_speech_timestamp = sub["end"]
        sub["words"] = words
    return last_speech_timestamp


def get_end(segments: List[dict]) -> Optional[float]:
    """end of the last word of the last segment that has words, else the last segment's end, else None"""
    for seg in reversed(segments):
        if seg["words"]:
            return seg["words"][-1]["end"]
    return segments[-1]["end"] if segments else None


def restore_speech_timestamps(segments: Iterable, speech_chunks: List[dict], sampling_rate: int) -> Iterable:
    """Maps segment / word times from the silence-free axis back onto the recording (generator)."""
    ts_map = SpeechTimestampsMap(speech_chunks, sampling_rate)
    for seg in segments:
        if seg.words:
            for w in seg.words:
                # both ends of a word resolve in the chunk that contains its middle
                k = ts_map.get_chunk_index((w.start + w.end) / 2)
                w.start = ts_map.get_original_time(w.start, k)
                w.end = ts_map.get_original_time(w.end, k)
            seg.start, seg.end = seg.words[0].start, seg.words[-1].end
        else:
            seg.start = ts_map.get_original_time(seg.start)
            seg.end = ts_map.get_original_time(seg.end, is_end=True)
        yield seg
