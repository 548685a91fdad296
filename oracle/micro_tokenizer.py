"""TEST INFRASTRUCTURE — a tiny in-memory `tokenizers.Tokenizer` with Whisper's vocabulary LAYOUT.

No Whisper `tokenizer.json` exists offline, so host-logic parity (prompts, word splitting, word timestamps,
the sequential seek loop) is exercised with a 400-entry byte-level BPE vocabulary followed by the special
tokens in Whisper's multilingual order — exactly the ids of `faster_whisper_amd.config.get_config("micro")`:
    0..255 bytes, 256..399 merges/padding, 400 <|endoftext|>, 401 <|startoftranscript|>, 402..405 languages
    (en zh de es), 406 <|translate|>, 407 <|transcribe|>, 408 <|startoflm|>, 409 <|startofprev|>,
    410 <|nospeech|>, 411 <|notimestamps|>, 412.. <|0.00|> ... <|30.00|>
The same object drives the REFERENCE's Tokenizer (oracle/gen_golden_host.py) and this repository's, so both
see identical ids.  Only tests/ and oracle/ import this module.
"""
from typing import List

WORDS = [" the", " and", " of", " to", " in", " is", " that", " it", " was", " for", " hello", " world", " whisper",
         " model", " speech", " audio", " time", " stamp", " word", " test", "ing", "ed", "er", "ly", " a", " I",
         " you", " we", " they", " not", " on", " with", " as", " be", " at", " this", " have", " from", " or", " one"]

N_TEXT = 400
LANGS = ["en", "zh", "de", "es"]


def _byte_alphabet() -> List[str]:
    from tokenizers import pre_tokenizers
    return sorted(pre_tokenizers.ByteLevel.alphabet())


def build():
    """-> tokenizers.Tokenizer (byte-level BPE + Whisper-style special tokens at the micro ids)"""
    import tokenizers
    from tokenizers import decoders, models, pre_tokenizers

    alphabet = _byte_alphabet()
    assert len(alphabet) == 256
    vocab = {ch: i for i, ch in enumerate(alphabet)}
    merges = []
    # byte -> unicode stand-in used by the byte-level pre-tokenizer (space becomes 'Ġ')
    probe = tokenizers.Tokenizer(models.BPE(vocab=dict(vocab), merges=[]))
    probe.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)

    def units(word):
        return [t for t in probe.encode(word, add_special_tokens=False).tokens]

    for w in WORDS:
        parts = units(w)
        while len(parts) > 1:
            # left-to-right pairwise merging; register every intermediate symbol once
            a, b = parts[0], parts[1]
            merged = a + b
            if merged not in vocab:
                if len(vocab) >= N_TEXT:
                    break
                vocab[merged] = len(vocab)
                merges.append((a, b))
            parts = [merged] + parts[2:]
    i = 0
    while len(vocab) < N_TEXT:                      # unreachable filler so that <|endoftext|> lands on id 400
        vocab[f"Āpad{i}"] = len(vocab)
        i += 1
    tok = tokenizers.Tokenizer(models.BPE(vocab=vocab, merges=merges))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    specials = (["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{c}|>" for c in LANGS]
                + ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>",
                   "<|notimestamps|>"] + [f"<|{i * 0.02:.2f}|>" for i in range(1501)])
    tok.add_special_tokens(specials)
    assert tok.token_to_id("<|endoftext|>") == 400 and tok.token_to_id("<|notimestamps|>") == 411
    assert tok.token_to_id("<|0.00|>") == 412 and tok.get_vocab_size() == 412 + 1501
    return tok
